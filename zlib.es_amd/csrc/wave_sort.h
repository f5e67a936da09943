// Bitonic sort of one 32-bit key per lane across a 64-lane wave (ascending by
// lane), in registers, for gfx950.  Every lane of the wave must be active.
//
// A compare-exchange with the lane at distance j is three VALU instructions where
// the partner is reachable by a single DPP pattern (j = 1, 2: quad_perm; j = 8:
// row_ror:8): a DPP min, a DPP max and a select.  j = 4 takes two banked row
// shifts, j = 16 / 32 the gfx950 row / half-wave swaps (v_permlane16_swap,
// v_permlane32_swap) and a select: no step goes through the LDS crossbar.  Which lanes keep the
// smaller key at a step is a compile-time 64-bit mask held in scalar registers.
#ifndef ZGPU_WAVE_SORT_H
#define ZGPU_WAVE_SORT_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zgpu_sort {

// lanes that keep the minimum at merge size k, distance j
constexpr uint64_t takemin_mask(uint32_t k, uint32_t j) {
    uint64_t m = 0;
    for (uint32_t l = 0; l < 64; ++l) {
        const bool up = (l & k) == 0, lower = (l & j) == 0;
        if (lower == up) m |= 1ull << l;
    }
    return m;
}

// key <- takemin ? min(key, partner) : max(key, partner), partner = DPP pattern of
// key (gfx9 DPP covers VOP1/VOP2, so min and max carry the pattern).  The s_nop
// covers the VALU-write -> DPP-read hazard.
#define ZGPU_CX_DPP(key, mask, CTRL)                                                  \
    do {                                                                              \
        uint32_t mn_, mx_;                                                            \
        asm volatile("s_nop 1\n\t"                                                    \
                     "v_min_u32_dpp %1, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
                     "v_max_u32_dpp %2, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
                     "v_cndmask_b32_e64 %0, %2, %1, %3"                               \
                     : "+v"(key), "=&v"(mn_), "=&v"(mx_)                              \
                     : "s"(mask));                                                    \
    } while (0)

__device__ inline uint32_t cx_other(uint32_t key, uint32_t other, uint64_t mask) {
    const uint32_t mn = key < other ? key : other, mx = key < other ? other : key;
    uint32_t out;
    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(out) : "v"(mx), "v"(mn), "s"(mask));
    return out;
}

template <uint32_t K, uint32_t J>
__device__ inline uint32_t cx(uint32_t key) {
    constexpr uint64_t mask = takemin_mask(K, J);
    if constexpr (J == 1) {
        ZGPU_CX_DPP(key, mask, "quad_perm:[1,0,3,2]");
    } else if constexpr (J == 2) {
        ZGPU_CX_DPP(key, mask, "quad_perm:[2,3,0,1]");
    } else if constexpr (J == 8) {
        ZGPU_CX_DPP(key, mask, "row_ror:8");
    } else if constexpr (J == 4) {
        // lanes 0-3 / 8-11 of a row read lane+4, lanes 4-7 / 12-15 read lane-4
        uint32_t o;
        asm volatile("s_nop 1\n\t"
                     "v_mov_b32_dpp %0, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                     "v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xa"
                     : "=&v"(o)
                     : "v"(key));
        key = cx_other(key, o, mask);
    } else {
        static_assert(J == 16 || J == 32, "distance");
        // gfx950 row / half-wave swaps of the key with itself leave the even rows (lower half)
        // in every row (half) of .x and the odd rows (upper half) in .y: a lane's partner is in
        // .y if the lane sits in an even row (lower half), else in .x
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        constexpr uint64_t upper = J == 16 ? 0xffff0000ffff0000ull : 0xffffffff00000000ull;
        v2u s;
        if constexpr (J == 16)
            s = __builtin_amdgcn_permlane16_swap(key, key, false, false);
        else
            s = __builtin_amdgcn_permlane32_swap(key, key, false, false);
        uint32_t other;
        asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(other) : "v"(s.y), "v"(s.x), "s"(upper));
        key = cx_other(key, other, mask);
    }
    return key;
}

template <uint32_t K, uint32_t J>
__device__ inline uint32_t merge_from(uint32_t key) {
    key = cx<K, J>(key);
    if constexpr (J > 1) key = merge_from<K, J / 2>(key);
    return key;
}

__device__ inline uint32_t wave_sort64(uint32_t key) {
    key = merge_from<2, 1>(key);
    key = merge_from<4, 2>(key);
    key = merge_from<8, 4>(key);
    key = merge_from<16, 8>(key);
    key = merge_from<32, 16>(key);
    key = merge_from<64, 32>(key);
    return key;
}

}  // namespace zgpu_sort
#endif
