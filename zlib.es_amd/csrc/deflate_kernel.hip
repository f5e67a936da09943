// DEFLATE encoder kernels for gfx950 (MI355X).
//
// Replace, per unit (block / zip entry / stream chunk), the reference's
// RawDeflate.compress() pipeline (src/rawdeflate.js:102-124): lz77 +
// searchLongestMatch_ (:283-395), the Huffman code construction (:474-605),
// the block header (:176-236) and the bit emission (:243-277,
// src/bitstream.js:42-95).
//
// Search kernel (zgpu_deflate_kernel, zgpu_deflate_thorough_kernel): one
// 1024-thread workgroup per CU, persistent over a unit queue.  The unit's
// input streams through a 64 KiB LDS ring (the 32 KiB window stays resident
// next to the hash heads and chain links).  The 16 waves take items of 64
// consecutive positions by ticket and carry each through
//   insert  4-byte hash per position, wave-wide sort of (hash, lane) pairs for
//           the predecessors inside the item, chains linked in item order
//   search  one lane per position walks its chain (bounded depth) comparing
//           up to 32 bytes per candidate; long matches then offer their
//           distance to the other lanes of the wave and are extended
//   parse   greedy / lazy successor per position, the positions the parse
//           reaches by pointer doubling from the entry position the previous
//           item hands over (the frontier, advanced under a try-lock)
//   tokens  16-bit token slots to global memory at the base the frontier
//           assigned, symbol histograms by LDS atomics
// and every 1024 items (64 KiB of input) a block's histogram and token count
// go to global memory.
//
// Emit kernel (zgpu_emit_kernel): one 256-thread workgroup per unit, eight per
// CU, block after block:
//   codes   length-limited Huffman code lengths + canonical codes (dynamic)
//   header  run-length coded code lengths and the code-length code
//   bits    per-token bit strings, prefix sum of bit lengths, LDS-staged
//           OR-merge, coalesced dword stores
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "zgpu_dev.h"
#include "wave_sort.h"

namespace {

// Optional per-phase cycle counters (diagnostic build only: -DZGPU_PROFILE).
#ifdef ZGPU_PROFILE
#define PH_DECL unsigned long long ph_t0 = 0, ph_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PH_START() do { if (tid == 0) ph_t0 = clock64(); } while (0)
#define PH_END(i) do { if (tid == 0) { unsigned long long t_ = clock64(); ph_acc[i] += t_ - ph_t0; ph_t0 = t_; } } while (0)
#define PH_FLUSH() do { if (tid == 0 && P.prof) for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&P.prof[i_], ph_acc[i_]); } while (0)
#else
#define PH_DECL
#define PH_START()
#define PH_END(i)
#define PH_FLUSH()
#endif

constexpr int WG = 1024;
constexpr int HBITS = 13;           // hash-head table entries = 1 << HBITS
constexpr uint32_t RING = 65536;    // LDS input ring (bytes)
constexpr uint32_t RING_PAD = 272;  // mirror of ring[0..272) after the end
constexpr uint32_t WMASK = 0x7fff;  // chain slots (32 Ki positions)
#ifndef ZGPU_LAG_FULL
#define ZGPU_LAG_FULL 54
#endif
#ifndef ZGPU_FAR_BLOCK
#define ZGPU_FAR_BLOCK 1
#endif
#ifndef ZGPU_PEND_FULL
#define ZGPU_PEND_FULL 3
#endif
constexpr uint32_t LAG = ZGPU_LAG_FULL;  // an item starts only when all items more than LAG back are parsed
constexpr uint32_t TAIL_MAXDIST = 16;  // tail walks start from matches at most this far back (runs, short periods)
constexpr int TAIL_STEPS = 8;         // candidates of a tail walk
#ifndef ZGPU_AHEAD
#define ZGPU_AHEAD 0
#endif
constexpr uint32_t AHEAD = ZGPU_AHEAD;         // the chains are linked this many items ahead of the search
constexpr uint32_t MAXDIST = 32768 - 64 * (LAG + AHEAD) - 128;  // window minus the span of in-flight items
constexpr uint32_t BLK_ITEMS = 1024;  // one DEFLATE block per 64 KiB of input
constexpr uint32_t LOOKAHEAD = 64 + 6 * 64 + 258 + 16;  // bytes an item may read past its start
constexpr uint32_t REFILL = 16384;  // ring refill granule
constexpr uint32_t CMP_CAP = 32;   // pass-1 compare length cap (bytes)
constexpr uint32_t SHARE_MIN = 16;  // matches at least this long offer their distance to the wave
constexpr int SHARE_ITERS = 6;      // distinct distances scored per wave ...
constexpr int SHARE_ITERS_THOROUGH = 16;  // ... and when the walk goes on past the first capped candidate


// ---- geometry of the search kernel ------------------------------------------------------------
// The constants above describe the full-window variant: one 1024-thread workgroup per CU with a
// 64 KiB ring and 32 Ki chain slots.  SearchCfg gathers what a variant may change; the kernel body
// reads its geometry from the variant's SearchCfg only.
#ifndef ZGPU_LAG_HALF
#define ZGPU_LAG_HALF 30
#endif
#ifndef ZGPU_LAG_QUARTER
#define ZGPU_LAG_QUARTER 12
#endif
struct CfgFull {  // 32 KiB window, one workgroup per CU
    static constexpr int WG = 1024;
    static constexpr int WG_PER_CU = 1;
    static constexpr uint32_t RING = 65536, CHAIN = 32768, HBITS = 13, NSLOT = 64, NSLOT_E = 128, LAG = ZGPU_LAG_FULL;
    static constexpr int PEND = ZGPU_PEND_FULL;
    // The ring is refilled an eighth at a time, so that it still holds the 32 KiB behind the oldest item in
    // flight while the newest bytes come in: matches reach as far back as the format allows.  (The chain
    // links of positions more than CHAIN - span of the items in flight back are already overwritten -- such a
    // position is still found as somebody's nearest earlier occurrence, but nothing beyond it.)
    static constexpr uint32_t REFILL = RING / 8;
    static constexpr bool FULL_DISTANCE = true;
};
#ifndef ZGPU_HALF_HBITS
#define ZGPU_HALF_HBITS 11
#define ZGPU_HALF_WGS 2
#endif
struct CfgHalf {  // units of at most 16 KiB: two workgroups per CU, 16 Ki chain slots
    static constexpr int WG = 1024;
    static constexpr int WG_PER_CU = ZGPU_HALF_WGS;
    static constexpr uint32_t RING = 32768, CHAIN = 16384, HBITS = ZGPU_HALF_HBITS, NSLOT = 32, NSLOT_E = 64, LAG = ZGPU_LAG_HALF;
    static constexpr int PEND = 2;  // 64 registers per lane: a third pending item would be paid in spills
    static constexpr uint32_t REFILL = RING / 4;
    static constexpr bool FULL_DISTANCE = false;  // the units fit the window as it is
};

struct CfgQuarter {  // units of at most 4 KiB: four workgroups of eight waves per CU, 8 Ki chain slots
    static constexpr int WG = 512;
    static constexpr int WG_PER_CU = 4;
    static constexpr uint32_t RING = 16384, CHAIN = 8192, HBITS = 10, NSLOT = 16, NSLOT_E = 32, LAG = ZGPU_LAG_QUARTER;
    static constexpr int PEND = 2;
    static constexpr uint32_t REFILL = RING / 4;
    static constexpr bool FULL_DISTANCE = false;
};

constexpr uint32_t huf_scratch_bytes(uint32_t cap) { return 17 * cap + 128; }  // see HufS

// ---- LDS map of the search kernel (bytes) ------------------------------------
constexpr uint32_t L_RING = 0;
constexpr uint32_t L_PREV = L_RING + RING + RING_PAD;      // u16[32768]
constexpr uint32_t L_HEAD = L_PREV + 65536;                // u16[1<<HBITS]
constexpr uint32_t L_FL = L_HEAD + (2u << HBITS);          // u32[288] lit/len freqs
constexpr uint32_t L_FD = L_FL + 288 * 4;                  // u32[32]  dist freqs
constexpr uint32_t L_MISC = L_FD + 32 * 4;                 // u32[64]  control words
constexpr uint32_t L_SCR = L_MISC + 64 * 4;                // parse bookkeeping of in-flight items
constexpr uint32_t NSLOT = 64;                       // > LAG
constexpr uint32_t S_EXIT = 0;                       // u16[NSLOT*64]
constexpr uint32_t S_CNT = S_EXIT + NSLOT * 64 * 2;  // u8[NSLOT*64]
constexpr uint32_t S_DONE = S_CNT + NSLOT * 64;      // u32[NSLOT]
// Entry position and token base of an item wait for the item's owner in a longer ring; the
// frontier stays less than NSLOT_E items ahead of every item whose tokens are not out yet.
constexpr uint32_t NSLOT_E = 128;
constexpr uint32_t S_ENT = S_DONE + NSLOT * 4;       // u32[NSLOT_E]
constexpr uint32_t S_TOKB = S_ENT + NSLOT_E * 4;     // u32[NSLOT_E]
constexpr uint32_t SCR_BYTES = S_TOKB + NSLOT_E * 4;
constexpr uint32_t L_TOTAL = L_SCR + SCR_BYTES;
static_assert(L_TOTAL <= 163840, "LDS budget");

// ---- LDS map of the emit kernel (bytes) --------------------------------------
constexpr uint32_t E_FL = 0;                        // u32[288] + u32[32] block histograms
constexpr uint32_t E_MISC = E_FL + 320 * 4;         // u32[64]
constexpr uint32_t E_LLTAB = E_MISC + 64 * 4;       // u32[288] code | len<<16
constexpr uint32_t E_DTAB = E_LLTAB + 288 * 4;      // u32[32]
constexpr uint32_t E_STAGE = E_DTAB + 32 * 4;       // u32[stage_words] bit staging
// staging words for nt threads per unit: header (160) + (4 nt / 2) tokens of 48 bits + carry, rounded up
constexpr int stage_words(int nt) { return (160 + (4 * nt / 2) * 48 / 32 + 2 + 127) / 128 * 128; }
static_assert(stage_words(256) == 1024, "staging capacity of the workgroup variant");
constexpr uint32_t E_HUF = E_STAGE + stage_words(256) * 4;  // Huffman build scratch (offset of the 256-thread variant)
// Huffman build scratch map (bytes from E_HUF): the lit/len tree, then the
// distance tree (reused for the code-length tree); once the lit/len and
// distance codes exist, the run-length items of the header reuse the first.
constexpr uint32_t H_TL = 0;
constexpr uint32_t H_TD = H_TL + huf_scratch_bytes(288);
constexpr uint32_t H_END = H_TD + huf_scratch_bytes(32);
constexpr uint32_t H_RLE = 0;      // u16[320] items: symbol | extra << 8
constexpr uint32_t H_CLF = 640;    // u32[32] code-length symbol freqs
constexpr uint32_t H_CLTAB = 768;  // u32[32]
constexpr uint32_t E_TOTAL = E_HUF + H_END;
constexpr uint32_t emit_lds_bytes(int nt) { return E_STAGE + (uint32_t)stage_words(nt) * 4 + H_END; }

// misc words (search kernel: M_NTOK, M_OLDEST, M_ITEM.., M_NEXT..; emit kernel: M_SCAN, M_UNIT, M_NTOK, M_TMP, M_OVF)
enum { M_SCAN = 0 /*[16]*/, M_UNIT = 32, M_NTOK = 34, M_TMP = 35, M_OVF = 36,
       M_ITEM = 40,    // next item to hand out
       M_LINK = 41,    // item whose turn it is to link its hash chains
       M_PARSE = 42,   // item whose turn it is to take the parse hand-off
       M_ENTRY = 43,   // unit position of the next token (belongs to item M_PARSE)
       M_LOADED = 44,  // input bytes resident in the ring (exclusive end)
       M_LOCK = 45,    // try-lock of the parse frontier
       M_SAFE = 46,    // the frontier may pass items below this without looking at M_OLDEST
       M_LINKED = 47,  // all items below this have written their chain links
       M_NEXT = 48,    // [10] the next unit: index, input offset (2), size, priming bytes, token offset (2), first block; tag; class
       M_OLDEST = 16   // [16] per wave: its oldest item whose tokens are not out yet (search kernel)
};

struct Lds {
    uint8_t *base;
    __device__ uint8_t *ring() const { return base + L_RING; }
    __device__ uint16_t *prev() const { return (uint16_t *)(base + L_PREV); }
    __device__ uint16_t *head() const { return (uint16_t *)(base + L_HEAD); }
    __device__ uint32_t *fl() const { return (uint32_t *)(base + L_FL); }
    __device__ uint32_t *fd() const { return (uint32_t *)(base + L_FD); }
    __device__ uint32_t *misc() const { return (uint32_t *)(base + L_MISC); }
    __device__ uint8_t *scr() const { return base + L_SCR; }
};

// 16 bytes at a global address of any alignment: the two aligned 16-byte pieces that hold them, shifted
// into place.  Both pieces contain bytes that were asked for, so nothing outside the pages of the
// caller's buffer is touched.  (Units of a ragged batch -- zip entries -- start at any byte.)
__device__ inline uint4 ldg16_any(const uint8_t *p) {
    const uint32_t a = (uint32_t)((uintptr_t)p & 15u);
    const uint4 *g = (const uint4 *)(p - a);
    const uint4 x = g[0];
    if (a == 0) return x;
    const uint4 y = g[1];
    const uint32_t r = a & 3u;
    uint32_t w0, w1, w2, w3, w4;
    switch (a >> 2) {
    case 0: w0 = x.x, w1 = x.y, w2 = x.z, w3 = x.w, w4 = y.x; break;
    case 1: w0 = x.y, w1 = x.z, w2 = x.w, w3 = y.x, w4 = y.y; break;
    case 2: w0 = x.z, w1 = x.w, w2 = y.x, w3 = y.y, w4 = y.z; break;
    default: w0 = x.w, w1 = y.x, w2 = y.y, w3 = y.z, w4 = y.w; break;
    }
    return make_uint4(__builtin_amdgcn_alignbyte(w1, w0, r), __builtin_amdgcn_alignbyte(w2, w1, r),
                      __builtin_amdgcn_alignbyte(w3, w2, r), __builtin_amdgcn_alignbyte(w4, w3, r));
}

// input bytes [lo, hi) of the unit -> ring (and the mirror of its first bytes);
// lo is a multiple of 16.  `nthreads` threads with ids `t` cooperate.  `aligned`: the unit starts at a
// multiple of 16 bytes (one load per 16 bytes instead of two and a shift).
template <uint32_t RING>
__device__ inline void load_ring(uint8_t *ring, const uint8_t *in, uint32_t lo, uint32_t hi, int t,
                                 int nthreads, bool aligned) {
    uint32_t done = lo;
    {
        const uint32_t hi16 = lo + ((hi - lo) & ~15u);
        for (uint32_t p = lo + (uint32_t)t * 16; p < hi16; p += (uint32_t)nthreads * 16) {
            const uint4 v = aligned ? *(const uint4 *)(in + p) : ldg16_any(in + p);
            const uint32_t ri = p & (RING - 1);
            *(uint4 *)(ring + ri) = v;
            if (ri < RING_PAD) *(uint4 *)(ring + RING + ri) = v;
        }
        done = hi16;
    }
    for (uint32_t p = done + (uint32_t)t; p < hi; p += (uint32_t)nthreads) {
        const uint32_t ri = p & (RING - 1);
        const uint8_t b = in[p];
        ring[ri] = b;
        if (ri < RING_PAD) ring[RING + ri] = b;
    }
}


// 16 bytes at any LDS address, and the number of leading bytes (0..16) in which
// two such pieces agree.  An LDS access that is not naturally aligned to its
// width is served one lane per cycle (64 cycles per wave instruction, measured:
// tools/lds_unaligned_bench.hip) against 7-11 cycles for aligned random reads,
// so the 16 bytes come as five aligned dwords and are shifted into place.
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ inline v4u ld16(const uint8_t *ring, uint32_t idx) {
    const uint32_t *w = (const uint32_t *)(ring + (idx & ~3u));
    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4];
    const uint32_t r = idx & 3u;
    return v4u{__builtin_amdgcn_alignbyte(d1, d0, r), __builtin_amdgcn_alignbyte(d2, d1, r),
               __builtin_amdgcn_alignbyte(d3, d2, r), __builtin_amdgcn_alignbyte(d4, d3, r)};
}
// index of the lowest set bit, 0xffffffff for 0 (one instruction, no zero test)
__device__ inline uint32_t ffbl(uint32_t x) {
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ inline uint32_t mlen16(v4u a, v4u b) {
    const v4u x = a ^ b;
    // straight-line: the first differing bit over the four words, 0xffffffff if none
    const uint32_t f0 = ffbl(x.x), f1 = ffbl(x.y) | 32u, f2 = ffbl(x.z) | 64u, f3 = ffbl(x.w) | 96u;
    uint32_t m = f0 < f1 ? f0 : f1;
    m = m < f2 ? m : f2;
    m = m < f3 ? m : f3;
    m >>= 3;
    return m < 16u ? m : 16u;
}

// Chains are keyed by four bytes although DEFLATE's shortest match is three: the candidates a
// walk of a few steps sees are then all worth comparing, which gives smaller output than keying
// by three bytes (on text by 2-4 %) and shorter walks; matches of exactly three bytes are only
// found where a four-byte chain happens to lead to them.
template <uint32_t HBITS>
__device__ inline uint32_t hash4(uint32_t w) {
    return (w * 0x9E3779B1u) >> (32 - HBITS);
}

// exclusive prefix sum over the workgroup; returns this thread's offset and
// the total through *total.  Uses misc[M_SCAN..+16].  Two barriers.
template <int NW>
__device__ inline uint32_t block_scan_excl(uint32_t v, uint32_t *total, uint32_t *part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) part[wave] = x;
    __syncthreads();
    uint32_t wsum = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        uint32_t p = part[w];
        if (w < wave) wsum += p;
        tot += p;
    }
    __syncthreads();
    *total = tot;
    return wsum + x - v;
}

// the same for NT threads that share a unit: one wave needs no barrier and no LDS
template <int NT>
__device__ inline uint32_t group_scan_excl(uint32_t v, uint32_t *total, uint32_t *part) {
    if constexpr (NT == 64) {
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t y = __shfl_up(x, d, 64);
            if ((threadIdx.x & 63) >= d) x += y;
        }
        *total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
        return x - v;
    } else {
        return block_scan_excl<NT / 64>(v, total, part);
    }
}

// Wave-to-wave hand-off through one LDS word.  The LDS executes a wave's
// instructions in issue order and is a single coherent memory for the
// workgroup, so a flag written after other LDS writes of the same wave lands
// after them, and a wave that has seen the flag may read what was written before
// it.  Only LDS is ordered here: no global-memory fence is involved.
// The words are accessed through explicit LDS-address-space pointers: a volatile
// access through a generic pointer would be issued as a (much slower) flat
// instruction.
typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
__device__ inline uint32_t lds_ld(const uint32_t *word) { return *(lds_vu32 *)word; }
__device__ inline void lds_st(uint32_t *word, uint32_t value) { *(lds_vu32 *)word = value; }
__device__ inline void lds_wait_eq(uint32_t *word, uint32_t want) {
    while (lds_ld(word) != want) __builtin_amdgcn_s_sleep(0);
    asm volatile("" ::: "memory");
}
__device__ inline void lds_wait_ge(uint32_t *word, uint32_t want) {
    while ((int32_t)(lds_ld(word) - want) < 0) __builtin_amdgcn_s_sleep(0);
    asm volatile("" ::: "memory");
}
__device__ inline void lds_publish(uint32_t *word, uint32_t value) {
    // The LDS executes a wave's instructions in issue order, so the flag lands after the
    // wave's earlier LDS writes without waiting for them; the compiler must keep that order.
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63) == 0) lds_st(word, value);
}

using zgpu_sort::wave_sort64;  // one key per lane, ascending by lane, in registers

// The neighbouring lane's value by a DPP wave shift (one VALU instruction, no LDS crossbar trip);
// every lane of the wave must be active.  from_below: lane i gets lane i - 1's value, lane 0 gets
// `edge`; from_above: lane i gets lane i + 1's, lane 63 gets `edge`.
__device__ inline uint32_t from_below(uint32_t v, uint32_t edge) {
    uint32_t r = edge;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v));
    return r;
}
__device__ inline uint32_t from_above(uint32_t v, uint32_t edge) {
    uint32_t r = edge;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v));
    return r;
}

// RFC 1951 length symbol for l = len-3 (0..255): symbol, extra bit count, extra value
__device__ inline void len_sym(uint32_t l, uint32_t &sym, uint32_t &eb, uint32_t &ev) {
    if (l < 8) {
        sym = 257 + l;
        eb = 0;
        ev = 0;
    } else if (l == 255) {
        sym = 285;
        eb = 0;
        ev = 0;
    } else {
        uint32_t n = 31 - __builtin_clz(l);
        sym = 257 + 4 * (n - 1) + ((l >> (n - 2)) & 3);
        eb = n - 2;
        ev = l & ((1u << eb) - 1);
    }
}
// distance symbol for dm = dist-1 (0..32767)
__device__ inline void dist_sym(uint32_t dm, uint32_t &sym, uint32_t &eb, uint32_t &ev) {
    if (dm < 4) {
        sym = dm;
        eb = 0;
        ev = 0;
    } else {
        uint32_t n = 31 - __builtin_clz(dm);
        sym = 2 * n + ((dm >> (n - 1)) & 1);
        eb = n - 1;
        ev = dm & ((1u << eb) - 1);
    }
}

// ---- serial bit writer into the staging words (thread 0 only) --------------
struct BitW {
    uint32_t *stage;
    uint32_t rel;  // bit position relative to stage[0]
    __device__ void put(uint32_t v, uint32_t nb) {
        if (!nb) return;
        uint32_t w = rel >> 5, sh = rel & 31;
        uint64_t x = (uint64_t)v << sh;
        stage[w] |= (uint32_t)x;
        if (sh + nb > 32) stage[w + 1] |= (uint32_t)(x >> 32);
        rel += nb;
    }
};

// ---- Huffman code lengths (length-limited), canonical codes ----------------
// One wave builds the code of one alphabet; the control flow is wave-uniform
// and the serial part (the two-queue merge) runs on scalar registers.
__device__ inline uint32_t rl(uint32_t v, uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
__device__ inline uint32_t wl(uint32_t old, uint32_t val, uint32_t l) {
    return (threadIdx.x & 63u) == l ? val : old;  // val and l are wave-uniform
}

struct HufS {          // LDS scratch of one tree, cap = alphabet size rounded up to 32
    uint32_t *sorted;  // [cap] keys (weight << 9 | symbol), ascending; m of them are real
    uint32_t *niw;     // [cap] weights of the internal nodes (sort keys before that)
    uint16_t *parN;    // [cap] parent of internal node
    uint16_t *parL;    // [cap] parent of leaf (by sorted index)
    uint16_t *dep;     // [cap] depth of internal node
    uint32_t *blc;     // [32]  symbols per code length
    uint8_t *lens;     // [cap] code length per symbol
    uint32_t *rank;    // [cap] sort ranks (aliases parN and the two bytes per entry before it)
};
__device__ inline HufS huf_scratch(uint8_t *base, uint32_t cap) {
    HufS S;
    S.sorted = (uint32_t *)base;
    S.niw = (uint32_t *)(base + 4 * cap);
    S.parN = (uint16_t *)(base + 10 * cap);
    S.parL = (uint16_t *)(base + 12 * cap);
    S.dep = (uint16_t *)(base + 14 * cap);
    S.blc = (uint32_t *)(base + 16 * cap);
    S.lens = base + 16 * cap + 128;
    S.rank = (uint32_t *)(base + 8 * cap);
    return S;
}

// S.sorted[0..m) holds the used symbols' keys in ascending order, m >= 2.
// Writes tab[s] = bit-reversed canonical code | length << 16 (0 for unused
// symbols) for s < nsym.  NCH = ceil(alphabet / 64).
template <int NCH>
__device__ void wave_huffman(const HufS &S, uint32_t m_in, uint32_t nsym, uint32_t maxbits, uint32_t *tab) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)m_in);
    const uint32_t INF = 0xffffffffu;
    for (uint32_t sy = lane; sy < nsym; sy += 64) S.lens[sy] = 0;
    if (lane < 32) S.blc[lane] = 0;
    // ---- two-queue merge, many steps per round.  Huffman's rule -- join the two lightest, leaves
    // before nodes of equal weight, nodes in order of creation -- consumes leaves and nodes each in
    // order.  Let t be the weight of the next node (the sum of the two lightest): every leaf and
    // node that weighs at most t is consumed before any node made from now on (those weigh at
    // least t and queue behind), so these items, in merged order, pair up (0,1), (2,3), ... exactly
    // as the serial rule would pair them, and all the pairs of a round are independent.  A round
    // looks at the next 32 leaves (lanes 0..31, ascending) and the next 32 nodes (lanes 63..32,
    // i.e. descending by lane: together a bitonic sequence, sorted by one merge network), takes
    // the prefix that weighs at most t, cuts it where a window ran out although its queue goes
    // on (items behind the window could belong in between), and joins the pairs.  Keys are
    // weight << 7 | tag with tag = window index for leaves and 64 + window index for nodes, which
    // orders equal weights as the rule asks.
    {
        uint32_t li = 0, ii = 0, k = 0;  // leaves consumed, nodes consumed, nodes made (wave-uniform)
        while (k + 1 < m) {
            const uint32_t j = lane < 32 ? lane : 63 - lane;
            uint32_t key = INF;
            if (lane < 32) {
                if (li + j < m) key = ((S.sorted[li + j] >> 9) << 7) | j;
            } else {
                if (ii + j < k) key = (S.niw[ii + j] << 7) | (64 + j);
            }
            key = zgpu_sort::merge_from<64, 32>(key);
            const uint32_t w = key >> 7;
            const uint32_t t = rl(w, 0) + rl(w, 1);  // at least two items are left while k + 1 < m
            const bool real = key != INF, is_node = (key & 64u) != 0;
            uint32_t c = (uint32_t)__builtin_popcountll(__ballot(real && w <= t));
            if (li + 32 < m) {
                const uint32_t cut = 64 - (uint32_t)__builtin_clzll(__ballot(real && !is_node) | 1ull);
                c = c < cut ? c : cut;
            }
            if (ii + 32 < k) {
                const uint32_t cut = 64 - (uint32_t)__builtin_clzll(__ballot(real && is_node) | 1ull);
                c = c < cut ? c : cut;
            }
            c &= ~1u;
            const bool in = lane < c;
            const uint32_t partner = __shfl_xor(w, 1, 64);
            const uint32_t q = k + (lane >> 1);
            if (in) {
                if (!(lane & 1u)) {
                    S.niw[q] = w + partner;
                    S.dep[q] = 1;
                }
                if (is_node)
                    S.parN[ii + (key & 31u)] = (uint16_t)q;
                else
                    S.parL[li + (key & 31u)] = (uint16_t)q;
            }
            const uint32_t nn = (uint32_t)__builtin_popcountll(__ballot(in && is_node));
            li += c - nn;
            ii += nn;
            k += c >> 1;
            asm volatile("" ::: "memory");  // the next round reads the nodes written here
        }
    }
    if (lane == 0) {
        S.parN[m - 2] = (uint16_t)(m - 2);  // the root
        S.dep[m - 2] = 0;
    }
    // ---- depth of every internal node by pointer jumping (reads of a round precede its writes)
    for (int round = 0; round < 10; ++round) {
        uint32_t np[NCH], nd[NCH];
        bool more = false;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint32_t k = (uint32_t)c * 64 + lane;
            np[c] = 0;
            nd[c] = 0;
            if (k + 1 < m) {
                const uint32_t pp = S.parN[k];
                np[c] = S.parN[pp];
                nd[c] = (uint32_t)S.dep[k] + S.dep[pp];
                more = more || np[c] != m - 2;
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint32_t k = (uint32_t)c * 64 + lane;
            if (k + 1 < m) {
                S.parN[k] = (uint16_t)np[c];
                S.dep[k] = (uint16_t)nd[c];
            }
        }
        asm volatile("" ::: "memory");
        if (!__ballot(more)) break;
    }
    // ---- symbols per code length; depths beyond the limit are clamped and repaired
    uint32_t overflow = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t i = (uint32_t)c * 64 + lane;
        bool over = false;
        if (i < m) {
            uint32_t d = (uint32_t)S.dep[S.parL[i]] + 1;
            over = d > maxbits;
            if (over) d = maxbits;
            atomicAdd(&S.blc[d], 1u);
        }
        overflow += (uint32_t)__builtin_popcountll(__ballot(over));
    }
    if (overflow && lane == 0) {
        // Clamping leaves the code over-subscribed by some number of codes of the maximum
        // length (how many depends on how deep the clamped leaves were, not only on their
        // count).  One step -- the deepest leaf above the limit moves one level down and a
        // leaf from the limit becomes its sibling -- frees exactly one such code.
        uint32_t kraft = 0;
        for (uint32_t b = 1; b <= maxbits; ++b) kraft += S.blc[b] << (maxbits - b);
        for (uint32_t excess = kraft - (1u << maxbits); excess > 0; --excess) {
            uint32_t bits = maxbits - 1;
            while (S.blc[bits] == 0) --bits;
            S.blc[bits]--;
            S.blc[bits + 1] += 2;
            S.blc[maxbits]--;
        }
    }
    asm volatile("" ::: "memory");
    const uint32_t blcv = lane < 32 ? S.blc[lane] : 0u;
    // ---- lengths in sorted order: the rarest symbols get the longest codes
    {
        uint32_t len[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) len[c] = 0;
        uint32_t acc = 0;
        for (uint32_t b = maxbits; b >= 1; --b) {
            const uint32_t nb = rl(blcv, b);
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (len[c] == 0 && (uint32_t)c * 64 + lane < acc + nb) len[c] = b;
            acc += nb;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint32_t i = (uint32_t)c * 64 + lane;
            if (i < m) S.lens[S.sorted[i] & 511u] = (uint8_t)len[c];
        }
    }
    // ---- canonical codes: first code per length, then symbols in index order
    uint32_t nextv = 0;
    {
        uint32_t code = 0;
        for (uint32_t b = 1; b <= maxbits; ++b) {
            code = (code + (b > 1 ? rl(blcv, b - 1) : 0u)) << 1;
            nextv = wl(nextv, code, b);
        }
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const uint32_t sy = (uint32_t)c * 64 + lane;
        const uint32_t l = sy < nsym ? (uint32_t)S.lens[sy] : 0u;
        uint32_t code = 0;
        for (uint32_t b = 1; b <= maxbits; ++b) {
            if (rl(blcv, b) == 0) continue;
            const unsigned long long mk = __ballot(l == b);
            const uint32_t base = rl(nextv, b);
            if (l == b) code = base + (uint32_t)__builtin_popcountll(mk & lt);
            nextv = wl(nextv, base + (uint32_t)__builtin_popcountll(mk), b);
        }
        if (sy < nsym) tab[sy] = l ? ((__brev(code) >> (32 - l)) | (l << 16)) : 0u;
    }
}

// ---- search kernel ----------------------------------------------------------
struct Params {
    const uint8_t *in;
    const uint64_t *in_off;     // n_units+1
    uint16_t *tok;              // token slots of this round
    const uint64_t *tok_off;    // n_units: first slot of the unit within tok
    const uint32_t *blk_first;  // n_units: index of the unit's first block
    uint32_t *hist;             // blocks x 320 symbol counts (lit/len 288, dist 32)
    uint32_t *ntok;             // blocks: tokens incl. end-of-block
    uint32_t *counter;          // unit queue head (relative to unit_first)
    uint32_t unit_first, unit_end;
    int lazy, max_chain, nice;
    int deep;                   // long-match steps: deep-walk budget (0 = off)
    const uint32_t *unit_list;  // NULL: units unit_first .. unit_end; else list_len unit numbers (and unit_end = ~0)
    uint32_t list_len;
    const uint32_t *unit_prime; // per unit: bytes before it (same stream, multiple of 64, <= 32 KiB) that prime its window
    const uint8_t *skip;        // NULL, or per unit: not zero = another launch searches this unit, pass it over
    uint32_t key3_max_bytes;    // first search: units of at most this many bytes are keyed on three bytes (0: none)
    const uint8_t *unit_class;  // NULL, or per unit (thorough variant): class 1..3 of a unit picks its chain budget
    int class_chain[3];         //   and its lazy value from these instead of max_chain and lazy
    int class_lazy[3];
    unsigned long long *prof;   // 16 phase counters (diagnostic builds) or null
};

// THOROUGH (chosen by nice_length > CMP_CAP): a candidate that reaches the compare
// cap is compared to its full length on the spot and the walk goes on until a
// candidate is nice_length long or the chain budget ends -- with nice_length 258 and
// a budget as long as the window that is the reference's exhaustive search
// (src/rawdeflate.js:370-395: longest wins, the nearest among equals).
// Thorough variant: positions that start eight or more bytes of one value are keyed on the value and the number
// of those bytes (see the insert step).
#ifndef ZGPU_RUN_KEY
#define ZGPU_RUN_KEY 1
#endif
// The same keys in the first search, for units that are not of the dense class but have a quarter of their bytes
// equal to their successor (indented text).  Measured and left off: of 115 units of indented JSON, 16 are above
// 1.03 x the reference's size under FIXED without and 12 with, the sums of six sets of real files gain 0.05 %,
// and the bench, none of whose units is flagged, loses 1.4 %.
#ifndef ZGPU_FIRST_RUN_KEY
#define ZGPU_FIRST_RUN_KEY 0
#endif
template <bool THOROUGH, class C>
__device__ __forceinline__ void search_units(const Params &P) {
    // the variant's geometry; these names shadow the full-window constants of the same name
    constexpr int WG = C::WG;
    constexpr uint32_t RING = C::RING, WMASK = C::CHAIN - 1, HBITS = C::HBITS;
    constexpr uint32_t NSLOT = C::NSLOT, NSLOT_E = C::NSLOT_E, LAG = C::LAG;
    // searched items a wave may hold before it must parse the oldest: the in-order hand-off waits for the slowest
    // item, and where items differ much in length (deep walks) a wave with room for a third goes on working
    // (bench with the deep walk on: 83.2 -> 81.9 ms; four are slower again)
    constexpr int PEND = C::PEND;
    constexpr uint32_t REFILL = C::REFILL;  // ring refill granule
    constexpr uint32_t LINKDIST = C::CHAIN - 64 * (LAG + AHEAD) - 128;  // chain slots minus the span of in-flight items
    // how far back a match may lie: the bytes must still be in the ring for the oldest item in flight
    constexpr uint32_t RINGDIST = RING - 64 * LAG - 2 * REFILL;
    constexpr uint32_t MAXDIST = !C::FULL_DISTANCE ? LINKDIST : RINGDIST < 32768u ? RINGDIST : 32768u;
    constexpr uint32_t L_RING = 0, L_PREV = L_RING + RING + RING_PAD, L_HEAD = L_PREV + 2 * C::CHAIN,
                       L_FL = L_HEAD + (2u << HBITS), L_FD = L_FL + 288 * 4, L_MISC = L_FD + 32 * 4,
                       L_SCR = L_MISC + 64 * 4;
    constexpr uint32_t S_EXIT = 0, S_CNT = S_EXIT + NSLOT * 64 * 2, S_DONE = S_CNT + NSLOT * 64,
                       S_ENT = S_DONE + NSLOT * 4, S_TOKB = S_ENT + NSLOT_E * 4, SCR_BYTES = S_TOKB + NSLOT_E * 4;
    constexpr uint32_t L_TOTAL = L_SCR + SCR_BYTES;
    static_assert(L_TOTAL * C::WG_PER_CU <= 163840, "LDS budget");
    static_assert(REFILL >= LOOKAHEAD && NSLOT > LAG && NSLOT_E > 2 * LAG, "geometry");
    // a refill may overwrite what lies a ring behind its end while the item LAG back still searches its window
    static_assert(MAXDIST + 64 * LAG + 2 * REFILL <= RING, "ring holds the window");
    static_assert(NSLOT > LAG && NSLOT_E > 2 * LAG, "parse-table rings hold the items in flight");
    __shared__ __align__(16) uint8_t lds_raw[L_TOTAL];
    const int tid = threadIdx.x, lane = tid & 63;
    uint8_t *ring = lds_raw + L_RING;
    uint16_t *prevt = (uint16_t *)(lds_raw + L_PREV);
    uint16_t *head = (uint16_t *)(lds_raw + L_HEAD);
    uint32_t *fl = (uint32_t *)(lds_raw + L_FL), *fd = (uint32_t *)(lds_raw + L_FD), *misc = (uint32_t *)(lds_raw + L_MISC);
    uint8_t *scr = lds_raw + L_SCR;
    PH_DECL;

    // The next unit's queue ticket and its entries of the unit tables are fetched by thread 0
    // while the current unit is being worked on, and wait in LDS (M_NEXT..): taking a unit then
    // costs no global round trips (they are three dependent ones: ticket, offsets, data, and
    // with one workgroup per CU nothing else would hide them), only the load of its first bytes.
    auto fetch_unit = [&](uint32_t tag) {
        // the queue hands out a range of units, or -- when a round's units are spread over kernel
        // variants by size -- the entries of a list
        uint32_t nu;
        for (;;) {
            const uint32_t ticket = atomicAdd(P.counter, 1u);
            nu = P.unit_list ? (ticket < P.list_len ? P.unit_list[ticket] : 0xffffffffu) : P.unit_first + ticket;
            if (nu >= P.unit_end || !P.skip || !(P.skip[nu] & 3u)) break;
        }
        uint64_t uo = 0, to = 0;
        uint32_t nn = 0, prm = 0, bf = 0;
        if (nu < P.unit_end) {
            uo = P.in_off[nu];
            nn = (uint32_t)(P.in_off[nu + 1] - uo);
            prm = nn ? P.unit_prime[nu] : 0u;
            to = P.tok_off[nu];
            bf = P.blk_first[nu];
        }
        misc[M_NEXT + 0] = nu;
        misc[M_NEXT + 1] = (uint32_t)uo;
        misc[M_NEXT + 2] = (uint32_t)(uo >> 32);
        misc[M_NEXT + 3] = nn;
        misc[M_NEXT + 4] = prm;
        misc[M_NEXT + 5] = (uint32_t)to;
        misc[M_NEXT + 6] = (uint32_t)(to >> 32);
        misc[M_NEXT + 7] = bf;
        if (THOROUGH) misc[M_NEXT + 9] = P.unit_class && nu < P.unit_end ? P.unit_class[nu] : 0u;
        else if (ZGPU_FIRST_RUN_KEY || C::WG_PER_CU > 1) misc[M_NEXT + 9] = P.skip && nu < P.unit_end ? P.skip[nu] : 0u;
        asm volatile("" ::: "memory");
        lds_st(&misc[M_NEXT + 8], tag);  // the entries above are complete (LDS stores land in order)
    };
    if (tid == 0) fetch_unit(0u);
    __syncthreads();
    // The next unit's first 16 KiB, 16 bytes per thread, requested when a wave runs out of
    // items in the current unit's last block and kept in registers until the unit changes.
    uint4 pf = make_uint4(0u, 0u, 0u, 0u);
    uint32_t pf_hi = 0;  // bytes [0, pf_hi) of the next unit are covered that way (uniform)
    for (;;) {
        // written before the barrier that ended the previous unit; overwritten only after the
        // barrier that ends this unit's init
        const uint32_t u = misc[M_NEXT + 0];
        if (u >= P.unit_end) break;

        // A unit that continues a stream starts `pr` bytes early: those bytes (the end of the
        // previous unit) go through the insert step only, so matches may reach back into them
        // as a decoder of the concatenated stream can; tokens start at position pr.
        const uint64_t uoff = (uint64_t)misc[M_NEXT + 1] | (uint64_t)misc[M_NEXT + 2] << 32;
        const uint32_t n_unit = misc[M_NEXT + 3];
        const uint32_t pr = misc[M_NEXT + 4];
        const uint8_t *in = P.in + uoff - pr;
        const uint32_t n = n_unit + pr;
        const uint32_t prime_items = pr >> 6;
        uint16_t *tokbuf = P.tok + ((uint64_t)misc[M_NEXT + 5] | (uint64_t)misc[M_NEXT + 6] << 32);  // of the block in progress
        const uint32_t unit_blk_first = misc[M_NEXT + 7];
        // chain budget and lazy value of this unit: the launch's, or (second search) those of the unit's class
        const uint32_t ubyte = THOROUGH || ZGPU_FIRST_RUN_KEY || C::WG_PER_CU > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)misc[M_NEXT + 9]) : 0u;
        // positions of runs are keyed on (value, run length): always in the thorough variant, else where the unit's class byte says so
        const bool run_keys = THOROUGH || (ZGPU_FIRST_RUN_KEY && (ubyte & 0x80u) != 0u);
        // A small unit under FIXED is keyed on three bytes like the reference's (src/rawdeflate.js:314-320): in a window
        // of a few KiB a chain is short however it is keyed, and the three-byte matches that a four-byte key hides are
        // worth 2-4 % of such a file (C headers cut to 1000 bytes: all of 16 above 1.03 x the reference's size, worst
        // 1.045 x).  In a full window the same key fills the walk's eight steps with short matches (+2 % on the bench
        // mix), and under DYNAMIC small files are 1-2 % smaller than the reference's without those matches.
        // (in the variants that serve small units: the full-window kernel is the same code with or without this)
        // (under DYNAMIC, P.key3_max_bytes is 0 and the unit's class byte asks for it: small units that look like the
        // tables of a binary -- see the classifier)
        const bool key3_first = !THOROUGH && C::WG_PER_CU > 1 && (n_unit <= P.key3_max_bytes || (ubyte & 0x08u) != 0u);
        const uint32_t first_key_mask = key3_first ? 0xffffffu : 0xffffffffu;
        const uint32_t min_found = key3_first ? 3u : 4u;
        const uint32_t ucls = ubyte & 3u;
        // bytes five to eight of a position that go into its chain key besides the first four (none, one, two, four)
        const uint32_t ukey = (ubyte >> 4) & 3u;
        const bool key3 = (ubyte & 0x40u) != 0u;  // the chains are keyed on three bytes
        const uint32_t key_mask = ukey == 0u ? 0u : ukey == 1u ? 0xffu : ukey == 2u ? 0xffffu : 0xffffffffu;
        const int u_chain = !THOROUGH || !ucls ? P.max_chain : ucls == 1u ? P.class_chain[0] : ucls == 2u ? P.class_chain[1] : P.class_chain[2];
        const int u_lazy = !THOROUGH || !ucls ? P.lazy : ucls == 1u ? P.class_lazy[0] : ucls == 2u ? P.class_lazy[1] : P.class_lazy[2];
        // Lower bounds of the parse frontier and of the ring's fill level that this wave has seen (both only
        // grow within a unit): an item whose bounds they already satisfy starts without reading the words --
        // two dependent LDS round trips fewer on the way to its linking turn.  Only in the variant with one
        // workgroup per CU: the others run at 64 registers per lane and pay for two more live values with
        // spills (zip workload +6 % search time when they carried them).
        constexpr bool CACHE_BOUNDS = C::WG_PER_CU == 1;
        uint32_t seen_parse = 0, seen_loaded = 0;

        // ---- unit init ----
        for (int i = tid; i < (1 << HBITS) / 8; i += WG)
            ((uint4 *)head)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
        for (int i = tid; i < 320; i += WG) fl[i] = 0;  // fl[288] + fd[32] are adjacent
        const uint32_t nitems = (n + 63) >> 6;  // items of 64 positions
        const bool aligned = (((uintptr_t)in) & 15u) == 0;
        PH_START();
        {
            // first 32 KiB of the unit -> ring (all threads)
            const uint32_t hi = n < 2 * REFILL ? n : 2 * REFILL;
            if ((uint32_t)tid * 16u < pf_hi) {
                *(uint4 *)(ring + tid * 16) = pf;
                if ((uint32_t)tid * 16u < RING_PAD) *(uint4 *)(ring + RING + tid * 16) = pf;
            }
            load_ring<RING>(ring, in, pf_hi, hi, tid, WG, aligned);
            pf_hi = 0;
            if (tid == 0) {
                misc[M_ITEM] = 0;
                misc[M_LINK] = 0;
                misc[M_LINKED] = 0;
                misc[M_PARSE] = 0;
                misc[M_ENTRY] = pr;
                misc[M_NTOK] = 0;
                misc[M_LOADED] = hi;
                misc[M_LOCK] = 0;
                misc[M_SAFE] = NSLOT_E;
            }
            if (tid < WG / 64) misc[M_OLDEST + tid] = 0;
            // stamps of the previous unit's items must not be taken for this unit's
            if (tid < (int)NSLOT) ((uint32_t *)(scr + S_DONE))[tid] = 0;
        }
        __syncthreads();
        PH_END(0);
        if (tid == 0) fetch_unit(u + 1);

        // ---- a block is done: end-of-block token, token count and histograms to global
        // memory for the emit kernel; the block state starts over ----
        auto flush_block = [&](uint32_t blk) {
            const uint32_t gb = unit_blk_first + blk;
            if (tid == 0) {
                const uint32_t nt = misc[M_NTOK];
                tokbuf[nt] = 256;
                P.ntok[gb] = nt + 1;
                misc[M_NTOK] = 0;
            }
            for (int i = tid; i < 320; i += WG) {  // a variant's workgroup may have fewer threads than counters
                P.hist[(size_t)gb * 320 + i] = i == 256 ? 1u : fl[i];  // fl[288] + fd[32] are adjacent
                fl[i] = 0;
            }
            __syncthreads();
        };
        if (n == 0) flush_block(0);
        // block b covers the unit's bytes [b * 64 KiB, ...): the priming items ride with block 0
        for (uint32_t blk = 0, blk_first = 0; blk_first < nitems; ++blk) {
            const uint32_t blk_cap = prime_items + (blk + 1) * BLK_ITEMS;
            const uint32_t blk_end = blk_cap < nitems ? blk_cap : nitems;
            // ---- item loop: every wave takes the next 64 positions, no workgroup barrier ----
            // The parse of an item is deferred until the wave has searched its next item, so
            // that the in-order parse hand-off rarely has to wait for a slower earlier item.
            // ---- parse bookkeeping shared by the waves ----
            // Per in-flight item (slot = item mod NSLOT): for every local index the exit
            // index and the number of tokens when the parse enters the item there, the
            // "searched" stamp, and -- once the parse frontier has passed the item -- its
            // entry position and token base.  Any wave moves the frontier (M_PARSE, M_ENTRY,
            // M_NTOK) under a try-lock; it never waits for the item's owner to search.
            // The owner reads entry and token base when it emits the item's tokens, which can be
            // a while after the frontier passed, so the frontier must not lap their ring before
            // that: every wave keeps the number of its oldest item whose tokens are not out in
            // M_OLDEST (it only grows), and the frontier stops NSLOT_E short of the smallest of
            // them.  It looks at the 16 words only when it reaches the bound it computed the
            // last time (M_SAFE), i.e. about once in NSLOT_E - 2 * LAG items.
            uint16_t *exit_tab = (uint16_t *)(scr + S_EXIT);
            uint8_t *cnt_tab = scr + S_CNT;
            uint32_t *slot_done = (uint32_t *)(scr + S_DONE);
            uint32_t *slot_entry = (uint32_t *)(scr + S_ENT);
            uint32_t *slot_tokb = (uint32_t *)(scr + S_TOKB);
            auto safe_bound = [&]() -> uint32_t {
                // four words at a time (this runs on whatever lanes the caller has active, often one; all the
                // words in registers at once would count against the variants that run at 64 VGPRs)
                uint32_t m = 0xffffffffu;
#pragma unroll 1
                for (int w = 0; w < WG / 64; w += 4) {
                    const uint32_t a = lds_ld(&misc[M_OLDEST + w]), b = lds_ld(&misc[M_OLDEST + w + 1]),
                                   c = lds_ld(&misc[M_OLDEST + w + 2]), d = lds_ld(&misc[M_OLDEST + w + 3]);
                    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d, x = ab < cd ? ab : cd;
                    m = x < m ? x : m;
                }
                return m + NSLOT_E;
            };
            // The frontier is one lane's serial work and every wave's start waits on it (LAG), so
            // its loop is kept to one LDS round trip per item: the stamp of the next item is
            // fetched together with the exit and count of the current one.  The LDS executes a
            // wave's instructions in order, so the stores that end a turn (state, M_PARSE,
            // unlock) need no waits between them.
            auto try_advance = [&]() {
                if (lane == 0) {
                    for (;;) {
                        if (atomicCAS(&misc[M_LOCK], 0u, 1u) != 0u) break;  // someone else is at it
                        // the frontier is one lane's serial walk that every wave's hand-offs hang on: while it
                        // runs, this wave's instructions go first among the four waves of its SIMD
                        __builtin_amdgcn_s_setprio(3);
                        uint32_t f = lds_ld(&misc[M_PARSE]), e = lds_ld(&misc[M_ENTRY]), t = lds_ld(&misc[M_NTOK]),
                                 safe = lds_ld(&misc[M_SAFE]);
                        uint32_t stamp = lds_ld(&slot_done[f & (NSLOT - 1)]);
                        bool held = false;  // stopped by the bound, not by an unsearched item
                        while (stamp == f + 1) {
                            if (f == safe) {
                                safe = safe_bound();
                                if (f == safe) {
                                    held = true;
                                    break;
                                }
                            }
                            const uint32_t sl = f & (NSLOT - 1), le = e - (f << 6);
                            slot_entry[f & (NSLOT_E - 1)] = e;
                            slot_tokb[f & (NSLOT_E - 1)] = t;
                            stamp = lds_ld(&slot_done[(f + 1) & (NSLOT - 1)]);
                            if (le < 64u) {
                                e = (f << 6) + exit_tab[sl * 64 + le];
                                t += cnt_tab[sl * 64 + le];
                            }
                            ++f;
                        }
                        lds_st(&misc[M_ENTRY], e);
                        lds_st(&misc[M_NTOK], t);
                        lds_st(&misc[M_SAFE], safe);
                        lds_st(&misc[M_PARSE], f);
                        lds_st(&misc[M_LOCK], 0u);
                        __builtin_amdgcn_s_setprio(0);
                        // an item stamped while the lock was held must not be left behind; nor a
                        // bound that moved (a wave that moves it after this look finds M_PARSE
                        // at the bound and advances the frontier itself)
                        if (lds_ld(&slot_done[f & (NSLOT - 1)]) != f + 1) break;
                        if (held && safe_bound() == f) break;
                    }
                }
            };
            // Up to PEND searched items wait per wave until the frontier has passed them
            // (slot 0 is the oldest); then their tokens are emitted.
            int npend = 0;
            // item; per lane: the position's byte << 24 | best length << 15 | best distance - 1
            uint32_t q_k[PEND], q_ld[PEND];
            // per lane: the item's positions on the parse path that starts at this lane (bit i: local index i)
            uint32_t q_rlo[PEND], q_rhi[PEND];
            // kind / successor of every local index of an item (0 literal, 1 match here,
            // 2 literal + match at +1); no look-ahead across the item boundary
            auto successor = [&](uint32_t best_len, uint32_t l1, uint32_t &kind) -> uint32_t {
                // l1: best length of the next position (0 for the item's last lane)
                uint32_t adv = 1;
                kind = 0;
                if (best_len >= 3) {
                    if (u_lazy > 0 && best_len < (uint32_t)u_lazy && l1 > best_len) {
                        kind = 2;
                        adv = 1 + l1;
                    } else {
                        kind = 1;
                        adv = best_len;
                    }
                }
                return (uint32_t)lane + adv;
            };
            // cur: the item the wave is working on (the block's end when it has taken its last)
            // must: wait for the frontier if it has not passed the item yet; else leave the item
            // waiting and return false.  One LDS round trip tells both whether it has passed
            // and, if so, where the item's tokens start.
            auto emit_oldest = [&](uint32_t cur, bool must) -> bool {
                // ---- tokens of item q_k[0]: lane l <-> local index l ----
                // The positions the parse visits in the item were collected per starting index
                // when the item was prepared; the set for the actual entry index comes by two
                // readlanes.  That path mask gives every lane its place in the token stream by
                // bit counts; no shuffles.
                const uint32_t k0 = q_k[0], ld = q_ld[0];
                const uint32_t p0 = k0 << 6, sl = k0 & (NSLOT_E - 1);
                const uint32_t seg_end = (n - p0) < 64u ? (n - p0) : 64u;
                PH_START();
                // the frontier writes an item's slot before it publishes M_PARSE, and the LDS
                // serves this wave's reads in order: slot values read after a M_PARSE that shows
                // the item passed are the item's
                uint32_t fr = lds_ld(&misc[M_PARSE]);
                uint32_t s_ent = lds_ld(&slot_entry[sl]), s_tok = lds_ld(&slot_tokb[sl]);
                while ((int32_t)(fr - k0) <= 0) {
                    if (!must) return false;
                    try_advance();
                    __builtin_amdgcn_s_sleep(1);
                    fr = lds_ld(&misc[M_PARSE]);
                    s_ent = lds_ld(&slot_entry[sl]);
                    s_tok = lds_ld(&slot_tokb[sl]);
                }
                asm volatile("" ::: "memory");
                if constexpr (CACHE_BOUNDS) seen_parse = (uint32_t)__builtin_amdgcn_readfirstlane((int)fr);
                // tokens out = this wave's oldest-item word moves on, which is what the frontier's bound
                // waits for: slightly ahead of the bulk search work of the SIMD's other waves
                __builtin_amdgcn_s_setprio(1);
                PH_END(15);
                // local index of the first token (>= seg_end: a match from before covers the item)
                const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ent) - p0;
                const uint32_t tokbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_tok);
                // both are in registers: this wave's oldest item with tokens to come is the next
                // one.  A frontier held at this item's bound either shows in M_PARSE now, or
                // its wave has yet to publish M_PARSE and looks at the bound again after that.
                lds_publish(&misc[M_OLDEST + (tid >> 6)], npend > 1 ? q_k[1] : cur);
                const bool frontier_held = lds_ld(&misc[M_PARSE]) == k0 + NSLOT_E;
                const uint32_t ldn = from_above(ld, 0u);  // the next position's match (lazy tokens)
                uint32_t kind;
                (void)successor((ld >> 15) & 0x1ffu, (ldn >> 15) & 0x1ffu, kind);
                const unsigned long long valid = seg_end == 64u ? ~0ull : ((1ull << seg_end) - 1ull);
                const unsigned long long M = __ballot(kind != 0) & valid;
                // the positions the parse visits in the item: the set that prep built for the
                // lane at the entry index (two readlanes, no walk)
                unsigned long long path = 0;
                if (e < seg_end) path = ((unsigned long long)rl(q_rhi[0], e) << 32 | rl(q_rlo[0], e)) & valid;
                const unsigned long long two = path & __ballot(kind == 2);
                const bool on = (path >> lane) & 1ull;
                if (on) {
                    // slots before this lane's: one per path position, one more per match, one
                    // more per literal + match
                    const unsigned long long mt = path & M;
                    const uint32_t idx =
                        __builtin_amdgcn_mbcnt_hi((uint32_t)(path >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)path, 0u)) +
                        __builtin_amdgcn_mbcnt_hi((uint32_t)(mt >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mt, 0u)) +
                        __builtin_amdgcn_mbcnt_hi((uint32_t)(two >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)two, 0u));
                    uint16_t *dst = tokbuf + tokbase + idx;
                    if (kind != 1) {
                        const uint32_t b = ld >> 24;
                        *dst++ = (uint16_t)b;
                        atomicAdd(&fl[b], 1u);
                    }
                    if (kind != 0) {
                        const uint32_t t = kind == 1 ? ld : ldn;
                        const uint32_t tl = (t >> 15) & 0x1ffu, td = (t & 0x7fffu) + 1;
                        dst[0] = (uint16_t)(0x8000u | (tl - 3));
                        dst[1] = (uint16_t)(td - 1);
                        uint32_t sym, eb, ev;
                        len_sym(tl - 3, sym, eb, ev);
                        atomicAdd(&fl[sym], 1u);
                        dist_sym(td - 1, sym, eb, ev);
                        atomicAdd(&fd[sym], 1u);
                    }
                }
#pragma unroll
                for (int i = 0; i + 1 < PEND; ++i) {
                    q_k[i] = q_k[i + 1];
                    q_ld[i] = q_ld[i + 1];
                    q_rlo[i] = q_rlo[i + 1];
                    q_rhi[i] = q_rhi[i + 1];
                }
                --npend;
                PH_END(7);
                if (frontier_held) try_advance();
                return true;
            };
            for (;;) {
                // A ticket t links item t into the chains and searches item t - AHEAD: the
                // chains run AHEAD items ahead of the search, so a searched position may follow
                // the links of the positions up to 256 bytes after it (the long-match steps).
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(&misc[M_ITEM], 1u);
                t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                if (t >= blk_end + AHEAD) break;
                // From the ticket to the end of its linking turn an item is on the workgroup's one serial
                // chain: the turns are taken in ticket order, so whatever delays this wave on its way to the
                // turn (hash, sort, issue slots lost to the other three waves of its SIMD) delays every
                // item behind it.  The wave runs that stretch at raised priority and drops back for the
                // bulk of its work (walks, scoring, parse tables), which only its own item waits for.
                // Measured on the 64 KiB-unit bench: waiting for the turn 1.1 -> 0.3 cycles per input byte.
                __builtin_amdgcn_s_setprio(2);
                const bool has_search = t >= AHEAD;  // tickets below AHEAD (the unit's first) only link
                const uint32_t k = t - AHEAD;        // the searched item (valid when has_search)
                PH_START();
                if (has_search) {
                    if (npend == 0) {
                        // nothing older waits in this wave (its first item of the block)
                        lds_publish(&misc[M_OLDEST + (tid >> 6)], k);
                        try_advance();
                    }
                    // An unparsed item more than LAG items back must be parsed first, or the
                    // waves that wait for it below could never be released.
                    while (npend > 0 && k > q_k[0] + LAG) emit_oldest(k, true);
                    __builtin_amdgcn_s_setprio(2);
                    // bound the spread of in-flight items: everything more than LAG items back is
                    // parsed, so no wave still reads ring bytes or chain slots that get recycled
                    if constexpr (!CACHE_BOUNDS) {
                        if (k > LAG) lds_wait_ge(&misc[M_PARSE], k - LAG);
                    } else if (k > LAG && (int32_t)(seen_parse - (k - LAG)) < 0) {
                        uint32_t f;
                        while ((int32_t)((f = lds_ld(&misc[M_PARSE])) - (k - LAG)) < 0) __builtin_amdgcn_s_sleep(0);
                        asm volatile("" ::: "memory");
                        seen_parse = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
                    }
                }
                const uint32_t p0 = has_search ? k << 6 : 0u;
                const uint32_t p = p0 + (uint32_t)lane;
                // Ring refill every 16 KiB, by the wave that crosses the boundary -- but only after it has
                // stamped its item (end of this iteration): the bytes are for items a quarter of the ring ahead,
                // which cannot start before this item is parsed (LAG) and wait for M_LOADED in any case, while
                // a 16 KiB global load in front of the linking turn held up every item behind this one, and in
                // front of the search it held up the parse frontier.
                auto refill = [&]() {
                    if (has_search && k && (p0 & (REFILL - 1)) == 0) {
                        uint32_t lo = p0 + REFILL, hi = p0 + 2 * REFILL;
                        if (hi > n) hi = n;
                        if (lo < hi) {
                            load_ring<RING>(ring, in, lo, hi, lane, 64, aligned);
                            lds_publish(&misc[M_LOADED], hi);
                        }
                    }
                };
                {
                    uint32_t need = p0 + LOOKAHEAD;  // covers the linked item (AHEAD items on) as well
                    if (need > n) need = n;
                    if constexpr (!CACHE_BOUNDS) {
                        lds_wait_ge(&misc[M_LOADED], need);
                    } else if ((int32_t)(seen_loaded - need) < 0) {
                        uint32_t f;
                        while ((int32_t)((f = lds_ld(&misc[M_LOADED])) - need) < 0) __builtin_amdgcn_s_sleep(0);
                        asm volatile("" ::: "memory");
                        seen_loaded = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
                    }
                }
                PH_END(8);
                // this position's first 16 bytes: compared against candidates below
                const v4u own = ld16(ring, p & (RING - 1));
                // ---- insert item t: sort (hash, lane) in registers, link chains in item order ----
                if (t < nitems) {
                    const uint32_t ip0 = t << 6, ip = ip0 + (uint32_t)lane;
                    uint32_t key = 0xffffffffu;
                    {
                        const uint32_t *w = (const uint32_t *)(ring + (ip & (RING - 1) & ~3u));
                        uint32_t first4 = __builtin_amdgcn_alignbyte(w[1], w[0], ip & 3u);
                        bool run_key = false;
                        if (ZGPU_RUN_KEY && (THOROUGH || ZGPU_FIRST_RUN_KEY) && run_keys) {
                            // A position that starts eight or more bytes of one value is keyed on the value and on how
                            // many bytes of it there are (up to 258).  Keyed on its first bytes it shares a chain with
                            // every position of every run of the value, hundreds to thousands of them; of a run's
                            // positions, though, only the one with as much of the run in front of it can match past the
                            // run's end -- those nearer the end match less, those further from it as much as this
                            // position has of its run, which the position one byte back (scored for the whole item
                            // before the walks) gives as well.  Keyed this way the chain holds that one position of
                            // every earlier run that is long enough, and nothing else.
                            const uint32_t b = first4 & 0xffu;
                            const unsigned long long eq = __ballot(ip + 1u < n && ring[(ip + 1u) & (RING - 1)] == b);
                            const unsigned long long rest = ~(eq >> lane);  // zeros shifted in from the top end the count
                            const uint32_t ones = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
                            uint32_t r = ones + 1u;
                            if (eq >> 63) {
                                // the item's last byte goes on into the next ones: how far, for the lanes it reaches
                                uint32_t ext = 0;
                                for (uint32_t q = 0; q < 4u; ++q) {
                                    const uint32_t jx = ip0 + 64u + (q << 6) + (uint32_t)lane;
                                    const unsigned long long off = ~__ballot(jx < n && ring[jx & (RING - 1)] == rl(b, 63));
                                    ext += off ? (uint32_t)__builtin_ctzll(off) : 64u;
                                    if (off) break;
                                }
                                if (ones == 64u - (uint32_t)lane) r = ones + ext;
                            }
                            r = r < 258u ? r : 258u;
                            if (r >= 8u && ip + 4u <= n) {
                                first4 = ((r << 8 | b) * 0x85EBCA6Bu) ^ 0x5bd1e995u;
                                run_key = true;
                            }
                        }
                        if constexpr (THOROUGH) {
                            // Units of a very few byte values (the dense class says how few): a chain keyed on four
                            // bytes holds a sixteenth of a two-valued unit's positions, and every one of them is a
                            // candidate the walk must at least test; keyed on eight, a 256th, and the matches that
                            // such data has are longer than eight bytes anyway.  The bytes past the unit's end are
                            // not part of a key.
                            // Units of many byte values of which a few weigh heavily (the tables of a binary: 16-bit
                            // values in 32-bit slots, small numbers among zeros) are keyed on three bytes, as the
                            // reference keys everything (src/rawdeflate.js:314-320): what repeats there is often three bytes long, and a chain
                            // keyed on four never shows it (of 256 units of shared libraries, 5 were above 1.03 x the
                            // reference's size under FIXED -- up to 1.09 x, at any chain budget -- and 1 is).
                            if (key3 && !run_key) first4 &= 0xffffffu;
                            if (key_mask && !run_key && ip + 4 <= n) {
                                const uint32_t left = n - ip - 4u;
                                uint32_t next4 = __builtin_amdgcn_alignbyte(w[2], w[1], ip & 3u) & key_mask;
                                if (left < 4u) next4 &= (1u << (left << 3)) - 1u;
                                first4 ^= next4 * 0x2545F491u;
                            }
                        }
                        if (!THOROUGH && C::WG_PER_CU > 1) first4 &= first_key_mask;
                        if (ip + 4 <= n) key = (hash4<HBITS>(first4) << 6) | (uint32_t)lane;  // all four bytes inside the unit
                    }
                    key = wave_sort64(key);
                    const uint32_t kp = from_below(key, 0u), kn = from_above(key, 0xffffffffu);
                    const uint32_t h = key >> 6;
                    const bool live = key != 0xffffffffu;
                    const bool first = lane == 0 || (kp >> 6) != h;
                    const bool last = lane == 63 || (kn >> 6) != h || kn == 0xffffffffu;
                    uint32_t mypos = ip0 + (key & 63u);
                    // Everything the turn needs is in registers before the wait, and its two
                    // accesses run on all lanes: a lane with no head to read reads any bucket and
                    // ignores the value; a lane with no head to write writes its position into
                    // its own link slot, which gets the real link right after the turn.
                    uint32_t pp = ip0 + (kp & 63u);
                    uint32_t roff = L_HEAD + 2u * (live ? h : 0u);
                    uint32_t woff = live && last ? roff : L_PREV + 2u * (mypos & WMASK);
                    asm volatile("" : "+v"(roff), "+v"(woff), "+v"(mypos));
                    PH_END(9);
                    lds_wait_eq(&misc[M_LINK], t);
                    PH_END(10);
                    // The turn is held only for the exchange on the bucket heads: the read of the
                    // old head and the write of the new one are issued in this order and the LDS
                    // executes them in this order, so the turn passes on without waiting for the
                    // read to come back.  The links that depend on it are written after that;
                    // M_LINKED (all items below it have written their links) orders those writes
                    // before any later item's walk.
                    const uint32_t hv = *(const uint16_t *)(lds_raw + roff);
                    *(uint16_t *)(lds_raw + woff) = (uint16_t)mypos;
                    lds_publish(&misc[M_LINK], t + 1);
                    __builtin_amdgcn_s_setprio(0);
                    if (first) pp = hv;
                    const uint32_t linked = lds_ld(&misc[M_LINKED]);
                    if (live) prevt[mypos & WMASK] = (uint16_t)pp;
                    if (linked != t) lds_wait_eq(&misc[M_LINKED], t);
                    lds_publish(&misc[M_LINKED], t + 1);
                } else {
                    // past the unit's last item: the items this search reads links of are all linked
                    lds_wait_ge(&misc[M_LINKED], nitems);
                    __builtin_amdgcn_s_setprio(0);
                }
                PH_END(11);
                if (!has_search) continue;
                // ---- search ----
                uint32_t best_len = 0, best_dist = 0;
                if (k >= prime_items) {  // priming items are only inserted
                    const uint32_t maxlen = p < n ? ((n - p) < 258u ? (n - p) : 258u) : 0u;
                    // how far back the links can be followed, and how far back a match may lie (the same number
                    // except in the variant whose ring holds more than its chain slots reach)
                    const uint32_t limit = p < LINKDIST ? p : LINKDIST;
                    const uint32_t flimit = p < MAXDIST ? p : MAXDIST;
                    const uint32_t pi = p & (RING - 1);
                    uint32_t blen = 2;
                    uint32_t c = 0, last = 0xffffu;  // the walk's next candidate and last distance (last = 0xffff: walk over)
                    // A distance scored for all 64 positions of the item at once: one byte compare
                    // per lane, a ballot, and count-trailing-ones give every lane its exact match
                    // length (up to 258) at that distance; a lane keeps it if it is longer than
                    // what it has (or as long and nearer).
                    auto score_distance = [&](uint32_t d) {
                        uint32_t run;
                        bool open;
                        {
                            const bool eq = p < n && p >= d && (own.x & 0xffu) == ring[(p - d) & (RING - 1)];
                            // zeros shifted in from the top read as mismatches after the inversion
                            const unsigned long long inv = ~(__ballot(eq) >> lane);
                            run = inv ? (uint32_t)__builtin_ctzll(inv) : 64u;
                            open = run == 64u - (uint32_t)lane;
                        }
                        // Beyond the item every match that is still open continues the same way:
                        // one number, how far the data at p0 + 64 goes on matching at this
                        // distance.  Four positions per lane and step: an aligned dword against
                        // a dword shifted into place, first differing byte by find-first-set,
                        // first differing lane by ballot.
                        if (__ballot(open)) {
                            // both steps' bytes are requested together: one LDS round trip
                            uint32_t xs[2];
#pragma unroll
                            for (uint32_t st = 0; st < 2; ++st) {
                                const uint32_t q4 = p0 + 64u + (st << 8) + ((uint32_t)lane << 2);
                                const uint32_t a = *(const uint32_t *)(ring + (q4 & (RING - 1)));
                                const uint32_t s4 = q4 - d;
                                const uint32_t b0 = *(const uint32_t *)(ring + (s4 & (RING - 1) & ~3u));
                                const uint32_t b1 = *(const uint32_t *)(ring + ((s4 + 4u) & (RING - 1) & ~3u));
                                uint32_t x = a ^ __builtin_amdgcn_alignbyte(b1, b0, s4 & 3u);
                                // positions past the unit's end differ by decree
                                const uint32_t left = q4 < n ? n - q4 : 0u;
                                if (left < 4u) x |= left ? 0xffffffffu << (left << 3) : 0xffffffffu;
                                xs[st] = x;
                            }
                            uint32_t ext = 512u;
                            const unsigned long long m0 = __ballot(xs[0] != 0u);
                            if (m0) {
                                const uint32_t fl_ = (uint32_t)__builtin_ctzll(m0);
                                ext = (fl_ << 2) + (rl(ffbl(xs[0]), fl_) >> 3);
                            } else {
                                const unsigned long long m1 = __ballot(xs[1] != 0u);
                                if (m1) {
                                    const uint32_t fl_ = (uint32_t)__builtin_ctzll(m1);
                                    ext = 256u + (fl_ << 2) + (rl(ffbl(xs[1]), fl_) >> 3);
                                }
                            }
                            if (open) run += ext;
                        }
                        if (run > maxlen) run = maxlen;
                        if (run >= 3 && d <= flimit && (run > blen || (run == blen && d < best_dist))) {
                            blen = run;
                            best_dist = d;
                        }
                    };
                    // An item whose 64 positions all repeat at one distance (long runs, long copies) takes
                    // that distance and skips the walks: tried only when the first and the last position's
                    // nearest candidates lie at the same distance, which the links alone tell.
                    bool whole = false;
                    if (p + 4 <= n) c = prevt[p & WMASK];
                    if (!THOROUGH && p0 + 67u <= n) {
                        const uint32_t d0 = (p0 - rl(c, 0)) & 0xffffu, d63 = (p0 + 63u - rl(c, 63)) & 0xffffu;
                        if (d0 == d63 && d0 >= 1u && d0 <= (p0 < MAXDIST ? p0 : MAXDIST)) {
                            const bool eq = (own.x & 0xffu) == ring[(p - d0) & (RING - 1)];
                            if (__ballot(eq) == ~0ull) {
                                score_distance(d0);
                                whole = true;
                            }
                        }
                    }
                    // Candidates per walk.  Where three quarters of the item's positions find their nearest candidate
                    // at the same distance as the position before them, the item lies in long copies (run-like data;
                    // text has a dozen such positions per item): the parse jumps from long match to long match -- the
                    // deep walk below sees to their length -- and what the positions in between would find further
                    // down their chains is hardly ever used.  Three candidates then, instead of max_chain
                    // (incrementing-run units: sizes +0.1 %).  Only with the deep walk on: without it those positions
                    // are where the parse lands after a match that was cut short.
                    int walk_budget = u_chain;
                    if (ZGPU_RUN_KEY && (THOROUGH || ZGPU_FIRST_RUN_KEY) && run_keys) {
                        // (see the chain keys of positions inside runs) the match one byte back, for the whole item
                        if (__ballot(p + 4 <= n && own.x == __builtin_amdgcn_alignbyte(own.x, own.x, 1u))) score_distance(1u);
                    }
                    if (!THOROUGH && !whole && P.deep > 0 && P.max_chain > 3) {
                        const uint32_t cb = from_below(c, 0xfffffffeu);
                        if (__builtin_popcountll(__ballot(p + 4 <= n && ((c - cb) & 0xffffu) == 1u)) >= 48) walk_budget = 3;
                    }
                    if (!whole) {
                        // pass 1, lane per position: walk the hash chain.  A candidate costs one
                        // LDS round trip: its chain link and its first 16 bytes (aligned dword
                        // reads, see ld16) are fetched together and compared against this
                        // position's 16 bytes held in registers, so there is no per-lane compare
                        // loop for the wave to diverge on.  Only a candidate that matches all 16
                        // bytes reads 16 more (CMP_CAP; pass 2 finds the exact length of long
                        // matches).
                        const uint32_t cmpmax = maxlen < CMP_CAP ? maxlen : CMP_CAP;
                        uint32_t own_tail = 0;  // THOROUGH: this position's four bytes that end at offset blen
                        if constexpr (THOROUGH && ZGPU_RUN_KEY) {
                            if (blen >= 4u && blen < maxlen) {  // (the match one byte back)
                                const uint32_t o = pi + blen - 3u;
                                const uint32_t *w = (const uint32_t *)(ring + (o & (RING - 1) & ~3u));
                                own_tail = __builtin_amdgcn_alignbyte(w[1], w[0], o & 3u);
                            }
                        }
                        // one candidate of this lane's walk; false: the walk is over
                        auto candidate = [&]() -> bool {
                            const uint32_t dist = (p - c) & 0xffffu;
                            if (dist <= last || dist > limit) {
                                last = 0xffffu;
                                return false;
                            }
                            last = dist;
                            const uint32_t ci = c & (RING - 1);
                            const uint32_t nxt = prevt[c & WMASK];
                            if (THOROUGH) {
                                // Only a longer match replaces the best so far (the walk goes from near to far and
                                // the nearest of equals wins), and a longer one agrees with this position in the
                                // four bytes that end at offset blen -- the reference's own first test
                                // (src/rawdeflate.js:377-380): two dwords tell, where the full compare reads five
                                // per 16 bytes of the best length.  Inside runs that is the difference between one
                                // step and sixteen per candidate (every earlier position of the run matches as far
                                // as the run goes and no further): a unit of runs of up to 128 bytes 130 -> 20 ms.
                                // Where the bytes are independent (two or three values) some lane of 64 passes the
                                // test at nearly every step and the wave pays for both paths: 25 -> 37 ms.
                                if (blen >= 4u) {
                                    const uint32_t o = ci + blen - 3u;
                                    const uint32_t *w = (const uint32_t *)(ring + (o & (RING - 1) & ~3u));
                                    if (__builtin_amdgcn_alignbyte(w[1], w[0], o & 3u) != own_tail) {
                                        c = nxt;
                                        return true;
                                    }
                                }
                            }
                            uint32_t len = mlen16(own, ld16(ring, ci));
                            if (len == 16 && cmpmax > 16)
                                len = 16 + mlen16(ld16(ring, pi + 16), ld16(ring, ci + 16));
                            if (THOROUGH) {
                                if (len == CMP_CAP) {
                                    while (len < maxlen) {
                                        const uint32_t m = mlen16(ld16(ring, (pi + len) & (RING - 1)),
                                                                  ld16(ring, (ci + len) & (RING - 1)));
                                        len += m;
                                        if (m < 16) break;
                                    }
                                }
                                if (len > maxlen) len = maxlen;
                            } else if (len > cmpmax) {
                                len = cmpmax;
                            }
                            c = nxt;
                            // selects, not a branch: every branch the wave can diverge on costs a
                            // handful of scalar instructions per candidate, and the scalar unit is
                            // shared by all waves of the CU (search kernel 75.3 -> 74.1 ms).  A loop
                            // with one exit per candidate -- the next link's range test folded into
                            // it -- was 2 ms slower than this one.
                            const bool better = len > blen;
                            blen = better ? len : blen;
                            best_dist = better ? dist : best_dist;
                            if (THOROUGH) {
                                if (better && blen >= 4u && blen < maxlen) {
                                    const uint32_t o = pi + blen - 3u;
                                    const uint32_t *w = (const uint32_t *)(ring + (o & (RING - 1) & ~3u));
                                    own_tail = __builtin_amdgcn_alignbyte(w[1], w[0], o & 3u);
                                }
                            }
                            // as long as the unit allows, or as long as the caller finds nice
                            // (nice <= CMP_CAP, the default: the first capped candidate ends
                            // the walk; pass 2 finds its exact length and the long-match
                            // steps below look for longer ones)
                            return !(better && (len >= maxlen || len >= (uint32_t)P.nice));
                        };
                        if (p + 4 <= n && !(THOROUGH && ZGPU_RUN_KEY && blen >= maxlen)) {
                            last = 0;
                            for (int budget = walk_budget; budget > 0; --budget) {
                                // two candidates in and not one position of the item (of those still walking) has
                                // found four bytes that repeat: incompressible data, whose chains hold hash
                                // collisions only -- the walk ends for the whole item (random units 56.6 -> 66.6
                                // GB/s; no named input and none of the bench's kinds changes size; ending after
                                // one candidate is no faster)
                                if (!THOROUGH && budget == walk_budget - 2 && !__ballot(blen >= min_found)) break;
                                if (!candidate()) break;
                            }
                            if constexpr (C::FULL_DISTANCE && ZGPU_FAR_BLOCK) {
                                // A walk that ran out of the links' reach stands at a candidate that is still worth a
                                // look: it is the nearest earlier occurrence behind the last one examined (that one's
                                // link slot is intact), and its bytes are in the ring.  Its own slot may already belong
                                // to a position one chain table further on, so nothing is followed from it.  This is
                                // what finds the copy of a file's preamble in the file before it, 29-32 KiB back.
                                // An item in the unit's first 29 KB has nothing that far behind it (a scalar test).  Else
                                // the candidate is looked at where a good part of the item's positions stand at one, not
                                // one or two: a copy that far back is found by several of them and shared with the others
                                // below, while text's scattered far candidates, looked at one by one, cost the walk 4 %
                                // for a tenth of a per cent of size.
                                if (p0 + 63u > LINKDIST) {
                                    const uint32_t fdist = (p - c) & 0xffffu;
                                    const bool fgo = fdist > limit && fdist <= flimit && blen < cmpmax;
                                    if (__builtin_popcountll(__ballot(fgo)) >= 16) {
                                        if (fgo) {
                                            const uint32_t ci = c & (RING - 1);
                                            uint32_t len = mlen16(own, ld16(ring, ci));
                                            if (len == 16 && cmpmax > 16) len = 16 + mlen16(ld16(ring, pi + 16), ld16(ring, ci + 16));
                                            if (len > cmpmax) len = cmpmax;
                                            if (len > blen) {
                                                blen = len;
                                                best_dist = fdist;
                                            }
                                        }
                                    }
                                }
                            }
                        }
                    }
                    PH_END(12);
                    // what follows the walk -- scoring, parse tables, tokens, the stamp -- runs one level above
                    // the walks of the SIMD's other waves: the sooner an item that is nearly done is stamped,
                    // the sooner the frontier moves (search 74.5 -> 73.5 ms; raising the walk's own later
                    // candidates above its earlier ones as well was 9 % slower)
                    __builtin_amdgcn_s_setprio(1);
                    if (!whole) {
                        // pass 2, distance sharing across the wave: a distance that gave one
                        // lane a long match is scored for all 64 positions at once.
                        // Offers: every lane's best distance if it is long enough to be worth
                        // sharing.
                        const uint32_t offer = best_dist;
                        constexpr int share_iters = THOROUGH ? SHARE_ITERS_THOROUGH : SHARE_ITERS;
                        unsigned long long todo = __ballot(blen >= SHARE_MIN);
                        for (int it = 0; it < share_iters; ++it) {
                            if (!todo) break;
                            const uint32_t d = rl(offer, (uint32_t)__builtin_ctzll(todo));  // a uniform lane: v_readlane, no LDS
                            score_distance(d);
                            todo &= ~__ballot(offer == d);
                        }
                        // Three-byte matches at the strides of small records.  The chains are keyed by four
                        // bytes, so a match of exactly three bytes is found only by accident -- and arrays of
                        // 32-bit numbers that differ in their low byte (symbol and hash tables, offsets,
                        // relocations) consist of nothing else: the reference codes each element as one literal
                        // and one three-byte match four bytes back (src/rawdeflate.js:370-395 keys on three
                        // bytes), which is a third less than four literals.  Whether the three bytes at a
                        // position repeat 4, 8 or 12 bytes further on is seen in the 16 bytes each lane holds;
                        // where enough positions without a match would gain one, that distance is scored for the
                        // whole item.
                        const uint32_t x4 = (own.x ^ own.y) << 8, x8 = (own.x ^ own.z) << 8, x12 = (own.x ^ own.w) << 8;
                        uint32_t xm = x4 < x8 ? x4 : x8;
                        xm = xm < x12 ? xm : x12;
                        if (__builtin_popcountll(__ballot(xm == 0u)) >= 6) {
                            const unsigned long long need = __ballot(p + 3u <= n && blen < 4u);
                            if (__builtin_popcountll((__ballot(x4 == 0u) << 4) & need) >= 4) score_distance(4u);
                            if (__builtin_popcountll((__ballot(x8 == 0u) << 8) & need) >= 4) score_distance(8u);
                            if (__builtin_popcountll((__ballot(x12 == 0u) << 12) & need) >= 4) score_distance(12u);
                        }
                    }
                    PH_END(13);
                    // ---- long matches: a lane whose match is long and does not just continue its
                    // predecessor's (same distance, one shorter) is where a parse may need something
                    // better than the nearest long candidate.  The reference looks at every earlier
                    // occurrence and keeps the longest (src/rawdeflate.js:370-395); two bounded
                    // steps go after the same matches here.
                    if (P.deep > 0 && __ballot(blen >= CMP_CAP && blen < maxlen)) {
                        // exact length (up to 258) of the match of position px at distance d, px and d
                        // uniform: four bytes per lane, first differing lane by ballot
                        auto coop_len = [&](uint32_t px, uint32_t d, uint32_t mx) -> uint32_t {
                            const uint32_t q4 = px + ((uint32_t)lane << 2), s4 = q4 - d;
                            const uint32_t *wa = (const uint32_t *)(ring + (q4 & (RING - 1) & ~3u));
                            const uint32_t *wb = (const uint32_t *)(ring + (s4 & (RING - 1) & ~3u));
                            // the ring's mirror covers a dword pair that straddles its end
                            uint32_t x = __builtin_amdgcn_alignbyte(wa[1], wa[0], q4 & 3u) ^
                                         __builtin_amdgcn_alignbyte(wb[1], wb[0], s4 & 3u);
                            const uint32_t off = (uint32_t)lane << 2;
                            const uint32_t left = off < mx ? mx - off : 0u;
                            if (left < 4u) x |= left ? 0xffffffffu << (left << 3) : 0xffffffffu;
                            const unsigned long long mm = __ballot(x != 0u);
                            if (mm) {
                                const uint32_t fl_ = (uint32_t)__builtin_ctzll(mm);
                                return (fl_ << 2) + (rl(ffbl(x), fl_) >> 3);
                            }
                            // 256 bytes agree; mx > 256 here: the last two (uniform addresses)
                            uint32_t l = 256;
                            while (l < mx && ring[(px + l) & (RING - 1)] == ring[(px + l - d) & (RING - 1)]) ++l;
                            return l;
                        };
                        const bool searchable = p + 4 <= n;
                        // The item's first position has no predecessor in the item; its match continues one from
                        // before when the byte in front of it repeats at the same distance (inside a long run or
                        // copy that is every item's first position: without this test each of them walks again
                        // what the position where the match began has walked already -- a third of all deep-walk
                        // candidates on incrementing-run data, four fifths on runs of equal bytes).
                        auto starts = [&]() -> unsigned long long {
                            const uint32_t pd = from_below(best_dist, 0u), pl = from_below(blen, 0u);
                            bool cont = pd == best_dist && pl == blen + 1u;
                            if (lane == 0 && blen >= CMP_CAP && p > best_dist)
                                cont = ring[(p - 1u) & (RING - 1)] == ring[(p - 1u - best_dist) & (RING - 1)];
                            return __ballot(searchable && blen >= CMP_CAP && blen < maxlen && !cont);
                        };
                        if (P.deep > 0) {
                            bool improved = false;  // this position's match grew in one of the walks below
                            // deep walk: the start lanes go on along their chains; a candidate is
                            // looked at only if the byte after the current best agrees, and then
                            // measured by the whole wave
                            unsigned long long sm = starts();
                            bool go = ((sm >> lane) & 1ull) && last != 0xffffu;
                            if (__ballot(go)) {
                                // one or two lanes' dependent chain with the whole wave behind it
                                __builtin_amdgcn_s_setprio(3);
                                // A match longer than the best one, at whatever distance, repeats the four
                                // bytes that straddle the best one's end (three inside, one after) at that
                                // distance: the candidate is on the chain of the position t that starts those
                                // four bytes.  When t's link leads to no earlier occurrence within reach,
                                // nothing longer exists and the walk ends (or never starts): a match that
                                // ends because the data goes on differently than anywhere before -- the
                                // usual case where runs and records of different lengths follow each other
                                // -- then costs two round trips instead of the whole chain.  (A hash
                                // collision in front of a real occurrence ends a walk too early; that costs
                                // size, never correctness.)  t lies up to 255 bytes on, in one of the next four
                                // items, and its link exists once that item is linked: those items are in other
                                // waves' hands or are taken by the waves that finish next, and linking an item
                                // waits for nothing this wave holds, so the wait below ends.  The horizon is a
                                // function of the item number alone: the output does not depend on timing.
                                uint32_t horizon = 0xffffffffu;
                                if constexpr (AHEAD == 0) {
                                    constexpr uint32_t AHEAD_ITEMS = WG / 64 < 5 ? WG / 64 : 5;
                                    const uint32_t hz = k + AHEAD_ITEMS < blk_end ? k + AHEAD_ITEMS : blk_end;
                                    lds_wait_ge(&misc[M_LINKED], hz);
                                    horizon = hz << 6;
                                }
                                PH_END(4);
                                // The walk itself is a loop of its own per lane with nothing in it but the
                                // chain: the next link and the candidate's byte where the best match ends,
                                // requested together (one round trip per candidate), until that byte agrees
                                // -- a candidate worth measuring -- or the chain, the window or the budget
                                // ends.  Everything that involves the whole wave (ballots, the measurement,
                                // the look at t) happens between such stretches, once per improvement and
                                // not once per candidate: the wave's instructions per candidate, not the
                                // round trip, were what a deep walk cost.
                                bool chk = go;
                                uint32_t tb = ring[(pi + blen) & (RING - 1)];
                                uint32_t seq = 0;
                                int budget = P.deep;
                                for (;;) {
                                    if (__ballot(chk)) {
                                        if (chk) {
                                            const uint32_t t = p + blen - 3u;
                                            if (t + 4u <= n && t < horizon) {
                                                const uint32_t c2 = prevt[t & WMASK];
                                                const uint32_t d = (t - c2) & 0xffffu;
                                                bool ok = d >= 1u && d <= limit;
                                                if (ok) {
                                                    const uint32_t s4 = t - d;
                                                    const uint32_t *wt = (const uint32_t *)(ring + (t & (RING - 1) & ~3u));
                                                    const uint32_t *ws = (const uint32_t *)(ring + (s4 & (RING - 1) & ~3u));
                                                    ok = __builtin_amdgcn_alignbyte(wt[1], wt[0], t & 3u) ==
                                                         __builtin_amdgcn_alignbyte(ws[1], ws[0], s4 & 3u);
                                                }
                                                go = go && ok;
                                            }
                                            chk = false;
                                        }
                                    }
                                    PH_END(5);
                                    bool hit = false;
                                    if (go) {
                                        // one exit, and selects instead of branches in the body: the loop
                                        // compiles to some twenty instructions around its two loads (a lane
                                        // whose chain has ended loads once more, from addresses masked into
                                        // range, and ignores the values)
                                        bool cont;
                                        do {
                                            const uint32_t dist = (p - c) & 0xffffu;
                                            // a chain that runs through consecutive positions is a run of one
                                            // period: nothing longer further down
                                            seq = dist == last + 1u ? seq + 1u : 0u;
                                            const bool ok = dist > last && dist <= limit && seq < 4u && budget > 0;
                                            const uint32_t cb = ring[(c + blen) & (RING - 1)];
                                            const uint32_t cn = prevt[c & WMASK];
                                            --budget;
                                            hit = ok && cb == tb;
                                            go = ok;
                                            last = ok ? dist : last;
                                            c = ok ? cn : c;
                                            cont = ok && !hit;
                                        } while (cont);
                                    }
                                    PH_END(6);
                                    unsigned long long hm = __ballot(hit);
                                    if (!hm) break;
                                    while (hm) {
                                        const uint32_t hl = (uint32_t)__builtin_ctzll(hm);
                                        hm &= hm - 1ull;
                                        const uint32_t px = p0 + hl, d = rl(last, hl), mx = rl(maxlen, hl);
                                        const uint32_t len = coop_len(px, d, mx);
                                        if ((uint32_t)lane == hl && len > blen) {
                                            blen = len;
                                            best_dist = d;
                                            tb = ring[(pi + blen) & (RING - 1)];
                                            if (len >= maxlen) go = false;
                                            chk = go;
                                            improved = true;
                                        }
                                    }
                                }
                            }
                            PH_END(2);
                            // Tail walk.  Where the match repeats at a short distance (a run, a short period) the
                            // position's own chain is no help: it runs through the run's every position.  The
                            // chain of t is: it holds the places where the run ended the same way before, each of
                            // which lines the match up at one distance d = t - candidate; the ones whose head
                            // agrees there (four bytes, and the byte after the best match) are measured (a hash
                            // collision on the chain is just one more candidate that fails this test).  The walk
                            // stays on that chain when the match grows -- the candidates further down line the
                            // same run end up with other continuations.  Start lanes here are also the item's
                            // first position and the first one whose match is shorter than the longest possible
                            // (inside a long run every position continues its predecessor's match, and the last 257
                            // of them are where a parse that lands there needs the match across the run's end).
                            // 4-value runs: 1.05-1.08 x the reference's size (FIXED) / 1.00-1.06 x (DYNAMIC) without
                            // it -> 1.00-1.02 x / 0.96-1.00 x.
                            {
                                const uint32_t pd = from_below(best_dist, 0u), pl = from_below(blen, 0u);
                                bool tgo = searchable && blen >= CMP_CAP && blen < maxlen && best_dist <= TAIL_MAXDIST &&
                                           !(lane != 0 && pd == best_dist && pl == blen + 1u && pl < 258u);
                                if (__ballot(tgo)) {
                                    __builtin_amdgcn_s_setprio(3);
                                    uint32_t horizon = 0xffffffffu;
                                    if constexpr (AHEAD == 0) {
                                        constexpr uint32_t AHEAD_ITEMS = WG / 64 < 5 ? WG / 64 : 5;
                                        const uint32_t hz = k + AHEAD_ITEMS < blk_end ? k + AHEAD_ITEMS : blk_end;
                                        lds_wait_ge(&misc[M_LINKED], hz);
                                        horizon = hz << 6;
                                    }
                                    auto ld4 = [&](uint32_t idx) -> uint32_t {  // four bytes at any ring position
                                        const uint32_t *w = (const uint32_t *)(ring + (idx & (RING - 1) & ~3u));
                                        return __builtin_amdgcn_alignbyte(w[1], w[0], idx & 3u);
                                    };
                                    const uint32_t t = p + blen - 3u;
                                    tgo = tgo && t + 4u <= n && t < horizon;
                                    uint32_t tc = prevt[t & WMASK];
                                    uint32_t tlast = 0;
                                    uint32_t tb = ring[(pi + blen) & (RING - 1)];
                                    int budget = TAIL_STEPS;
                                    for (;;) {
                                        bool hit = false;
                                        if (tgo) {
                                            bool cont;
                                            do {
                                                const uint32_t d = (t - tc) & 0xffffu;
                                                const bool ok = d > tlast && d <= limit && budget > 0;
                                                // (a lane whose chain has ended loads once more, from addresses
                                                // masked into range, and ignores the values)
                                                const uint32_t h4 = ld4(p - d);
                                                const uint32_t cb = ring[(p + blen - d) & (RING - 1)];
                                                const uint32_t cn = prevt[tc & WMASK];
                                                --budget;
                                                hit = ok && h4 == own.x && cb == tb;
                                                tgo = ok;
                                                tlast = ok ? d : tlast;
                                                tc = ok ? cn : tc;
                                                cont = ok && !hit;
                                            } while (cont);
                                        }
                                        unsigned long long hm = __ballot(hit);
                                        if (!hm) break;
                                        while (hm) {
                                            const uint32_t hl = (uint32_t)__builtin_ctzll(hm);
                                            hm &= hm - 1ull;
                                            const uint32_t px = p0 + hl, d = rl(tlast, hl), mx = rl(maxlen, hl);
                                            const uint32_t len = coop_len(px, d, mx);
                                            if ((uint32_t)lane == hl && len > blen) {
                                                blen = len;
                                                best_dist = d;
                                                tb = ring[(pi + blen) & (RING - 1)];
                                                if (len >= maxlen) tgo = false;
                                                improved = true;
                                            }
                                        }
                                    }
                                }
                            }
                            // what the walks found, for the lanes behind them
                            if (const unsigned long long im = __ballot(improved)) {
                                const uint32_t offer = best_dist;
                                unsigned long long todo = im;
                                for (int it = 0; it < 4 && todo; ++it) {
                                    const uint32_t d = rl(offer, (uint32_t)__builtin_ctzll(todo));
                                    score_distance(d);
                                    todo &= ~__ballot(offer == d);
                                }
                            }
                        }
                        __builtin_amdgcn_s_setprio(1);
                        PH_END(2);
                        PH_END(3);
                    }
                    if (blen >= 3 && !(blen == 3 && best_dist > 4096)) best_len = blen;
                }
                PH_END(13);
                PH_START();
                __builtin_amdgcn_s_setprio(1);
                uint32_t r_lo, r_hi;
                {
                    // ---- exit index and token count of every local index (six doubling steps) ----
                    const uint32_t seg_end = (n - p0) < 64u ? (n - p0) : 64u;
                    uint32_t kind;
                    const uint32_t l1 = from_above(best_len, 0u);
                    uint32_t J = successor(best_len, l1, kind);
                    if ((uint32_t)lane >= seg_end) J = (uint32_t)lane;  // past the end: stay put
                    uint32_t Cn = (uint32_t)lane < seg_end ? 1u + kind : 0u;  // slots: literal 1, match 2, literal + match 3
                    // and the positions visited on the way (this one included), for the token step
                    r_lo = (uint32_t)lane < seg_end && lane < 32 ? 1u << lane : 0u;
                    r_hi = (uint32_t)lane < seg_end && lane >= 32 ? 1u << (lane - 32) : 0u;
                    // an item without a single match (incompressible data): every index walks literal by
                    // literal to the item's end -- exit, count and visited set in closed form, no doubling
                    if (!__ballot(kind != 0u)) {
                        if ((uint32_t)lane < seg_end) {
                            const unsigned long long m = (seg_end == 64u ? ~0ull : (1ull << seg_end) - 1ull) & ~((1ull << lane) - 1ull);
                            J = seg_end;
                            Cn = seg_end - (uint32_t)lane;
                            r_lo = (uint32_t)m;
                            r_hi = (uint32_t)(m >> 32);
                        }
                    } else
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        // every index already leaves the item (long matches): nothing left to double
                        if (!__ballot(J < seg_end)) break;
                        // exit and count travel together: one permute per doubling step
                        const int src = (int)(J & 63u);
                        const uint32_t tt = (uint32_t)__shfl((int)(J | Cn << 16), src, 64);
                        const uint32_t tl = (uint32_t)__shfl((int)r_lo, src, 64), th = (uint32_t)__shfl((int)r_hi, src, 64);
                        const bool inb = J < seg_end;
                        Cn += inb ? tt >> 16 : 0u;
                        r_lo |= inb ? tl : 0u;
                        r_hi |= inb ? th : 0u;
                        J = inb ? (tt & 0xffffu) : J;
                    }
                    const uint32_t sl = k & (NSLOT - 1);
                    exit_tab[sl * 64 + lane] = (uint16_t)J;
                    cnt_tab[sl * 64 + lane] = (uint8_t)Cn;
                }
                PH_END(14);
                // tokens of items the frontier has passed leave before this item is stamped:
                // until then the frontier cannot run far enough to recycle their slots
                while (npend > 0 && emit_oldest(k, npend == PEND)) {
                }
#pragma unroll
                for (int i = 0; i < PEND; ++i)
                    if (i == npend) {
                        q_k[i] = k;
                        q_ld[i] = (own.x & 0xffu) << 24 | best_len << 15 | ((best_dist - 1) & 0x7fffu);
                        q_rlo[i] = r_lo;
                        q_rhi[i] = r_hi;
                    }
                ++npend;
                lds_publish(&slot_done[k & (NSLOT - 1)], k + 1);
                __builtin_amdgcn_s_setprio(0);
                // Only the wave whose item the frontier stands at has something to advance.  A
                // frontier on its way here either sees the stamp, or publishes M_PARSE = k after
                // this look and then looks at the stamp again (the check after its unlock).
                if (lds_ld(&misc[M_PARSE]) == k) try_advance();
                refill();
            }
            if (blk_end == nitems) {
                // out of items in the unit's last block: request the next unit's first bytes
                // now, so that they travel while the waves drain and meet at the barrier
                lds_wait_eq(&misc[M_NEXT + 8], u + 1);
                if (misc[M_NEXT + 0] < P.unit_end) {
                    const uint32_t npr = misc[M_NEXT + 4];
                    const uint8_t *nin = P.in + ((uint64_t)misc[M_NEXT + 1] | (uint64_t)misc[M_NEXT + 2] << 32) - npr;
                    const uint32_t nn = misc[M_NEXT + 3] + npr;
                    pf_hi = (nn < REFILL ? nn : REFILL) & ~15u;
                    if ((uint32_t)tid * 16u < pf_hi) pf = ldg16_any(nin + tid * 16);
                }
            }
            while (npend > 0) emit_oldest(blk_end, true);
            __builtin_amdgcn_s_setprio(0);
            // also a wave that got no item in this block must not hold the frontier back
            lds_publish(&misc[M_OLDEST + (tid >> 6)], blk_end);
            try_advance();
            PH_START();
            __syncthreads();
            PH_END(1);
            if (tid == 0) misc[M_ITEM] = blk_end + AHEAD;  // the counter overshoots by one per wave
            flush_block(blk);
            tokbuf += ZGPU_BLOCK_TOK;
            blk_first = blk_end;
        }

        PH_FLUSH();
#ifdef ZGPU_PROFILE
        for (int i_ = 0; i_ < 16; ++i_) ph_acc[i_] = 0;
#endif
    }
}

}  // namespace

extern "C" __global__ void __launch_bounds__(1024) zgpu_deflate_kernel(Params P) { search_units<false, CfgFull>(P); }
extern "C" __global__ void __launch_bounds__(1024) zgpu_deflate_thorough_kernel(Params P) { search_units<true, CfgFull>(P); }
// the occupancy experiment: half the window, two workgroups (32 waves) per CU, 64 registers per lane
extern "C" __global__ void __launch_bounds__(1024, 8) zgpu_deflate_half_kernel(Params P) { search_units<false, CfgHalf>(P); }
extern "C" __global__ void __launch_bounds__(512, 8) zgpu_deflate_quarter_kernel(Params P) { search_units<false, CfgQuarter>(P); }

namespace {

// ---- emit kernel ------------------------------------------------------------
// One 256-thread workgroup per unit (persistent over the unit queue) turns the
// token streams and histograms the search kernel left in global memory into
// DEFLATE blocks: Huffman codes (dynamic) or the fixed tables, the block header,
// and the token bits -- prefix sum of bit lengths, OR-merge into LDS staging
// words, coalesced dword stores.  Blocks of a unit follow each other at bit
// granularity, so a unit's blocks are emitted in order by one workgroup; many
// workgroups share a CU and hide each other's serial stretches.
// HCLEN order of the code-length symbols (RFC 1951 3.2.7): 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15
__device__ inline uint32_t hclen_order(uint32_t i) {
    const uint64_t oa = 0x0609070800121110ull;  // order[0..7], one byte each
    const uint64_t ob = 0x020d030c040b050aull;  // order[8..15]
    const uint32_t oc = 0x000f010eu;            // order[16..18]
    return i < 8 ? (uint32_t)(oa >> (8 * i)) & 0xffu : i < 16 ? (uint32_t)(ob >> (8 * (i - 8))) & 0xffu : i < 19 ? (oc >> (8 * (i - 16))) & 0xffu : 0u;
}

#ifndef ZGPU_EMIT_WAVES
#define ZGPU_EMIT_WAVES 8
#endif
struct EParams {
    const uint64_t *in_off;    // n_units+1 (unit sizes)
    uint8_t *out;              // direct: the caller's output buffer (4-byte aligned), units land at their final
                               // offsets; slots: a staging buffer, unit u goes to its bounded slot
    uint64_t out_cap;
    const uint64_t *slot_off;  // slots: n_units+1, 16-byte aligned
    uint64_t *out_off;         // n_total+1: exclusive prefix of the unit sizes, written here
    unsigned long long *state; // n_total look-back words, zero before the first round: flag << 62 | bytes
    uint32_t *out_size;        // n_units
    uint32_t *status;          // n_units: 0 ok, 1 emitted size differs from the computed one / slot too small, 2 out_cap too small
    const uint16_t *tok;       // token slots of this round
    const uint64_t *tok_off;   // n_units: first token of the unit within tok
    const uint32_t *blk_first; // n_units: index of the unit's first block
    uint32_t *hist;            // blocks x 320 symbol counts (lit/len 288, dist 32); a multi-block unit's code
                               // tables are parked here between its sizing and its emission
    const uint32_t *ntok;      // blocks: tokens incl. end-of-block
    uint32_t *counter;         // unit queue head (relative to unit_first)
    const uint8_t *unit_final; // n_units: 1 = the unit ends its stream (BFINAL), 0 = it ends with an empty stored block
    uint32_t unit_first, unit_end, n_total;
    const uint32_t *unit_list;  // NULL: units unit_first .. unit_end; else list_len unit numbers (and unit_end = ~0)
    uint32_t list_len;
    int ctype;
    unsigned long long *prof;  // diagnostic builds: 16 per-phase cycle sums of each unit's thread 0
};

// barrier of the NT threads that share a unit: the workgroup's, or -- one wave per unit -- none
// (a wave runs in lockstep and the LDS serves its requests in order; the compiler must keep them so)
template <int NT>
__device__ __forceinline__ void group_sync() {
    if constexpr (NT == 64)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else
        __syncthreads();
}

// Look-back over the units before u (decoupled: every unit first publishes its own size, then adds up
// its predecessors' until it meets one that already knows its inclusive prefix).  Called by one whole
// wave; returns the sum of the sizes of units [0, u).  A unit publishes before it waits and waits only
// for units whose tickets were drawn earlier, i.e. by workgroups that are running: no deadlock.  The wait
// is bounded all the same (a fraction of a second): ~0 comes back if a predecessor never showed up.
__device__ inline uint64_t lookback_offset(unsigned long long *state, uint32_t u, uint32_t bytes) {
    const int lane = threadIdx.x & 63;
    const unsigned long long VMASK = (1ull << 62) - 1ull;
    if (lane == 0)
        __hip_atomic_store(&state[u], ((u ? 1ull : 2ull) << 62) | bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t excl = 0;
    long long idx = (long long)u - 1;  // nearest unit not yet accounted for
    uint32_t spins = 0;
    while (idx >= 0) {
        const long long my = idx - lane;
        unsigned long long v = 2ull << 62;  // before unit 0: an inclusive prefix of zero
        if (my >= 0) v = __hip_atomic_load(&state[my], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t flag = (uint32_t)(v >> 62);
        const unsigned long long incl = __ballot(flag == 2), none = __ballot(flag == 0);
        const bool have = incl != 0;
        const int fi = have ? __builtin_ctzll(incl) : 63;  // nearest unit that knows its inclusive prefix
        const unsigned long long need = fi < 63 ? (1ull << (fi + 1)) - 1ull : ~0ull;  // lanes up to that unit
        if (none & need) {  // a unit on the way has not published yet
            if (++spins > (1u << 22)) return ~0ull;
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        unsigned long long c = ((need >> lane) & 1ull) ? (v & VMASK) : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
        excl += c;
        if (have) break;
        idx -= 64;
    }
    if (lane == 0 && u)
        __hip_atomic_store(&state[u], (2ull << 62) | (excl + bytes), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}

// NT threads per unit: 256 (a workgroup; units of any size) or 64 (one wave per unit, four units
// side by side in a workgroup: for small units the serial stretches of the code construction are
// what takes the time, and more of them then run at once).
//
// DIRECT: a unit is written straight to its final place in the caller's buffer: (1) the code tables
// and the header fields of its block(s) are built and the exact size follows from the histograms,
// (2) the look-back turns sizes into offsets, (3) header and tokens are emitted there.  A unit's
// first and last bytes may share a 32-bit word with its neighbours: those are stored bytewise.
// Otherwise (plans with rounds of small units: thousands of them are in flight at once and the
// look-back, a global round trip or several per unit, costs more than a gathering pass over the small
// output) a unit goes to its bounded slot of a staging buffer, block by block without the sizing
// step, and zgpu_scan_kernel / zgpu_compact_kernel place the units afterwards.
template <int NT, bool DIRECT>
__device__ __forceinline__ void emit_units(const EParams &P, uint8_t *lds) {
    constexpr int EWG = NT;                  // shadows the workgroup-wide constants above
    constexpr int ETOK = 4 * NT;
    constexpr int STAGE_WORDS = stage_words(NT);
    constexpr uint32_t E_HUF = E_STAGE + (uint32_t)STAGE_WORDS * 4;
    constexpr int M_META = 56;  // hlit, hdist, hclen, run-length items, header bits
    constexpr int M_EXCL = 38;  // [2] the unit's byte offset
    const int tid = threadIdx.x % NT, lane = tid & 63, wave = tid >> 6;
    uint32_t *fl = (uint32_t *)(lds + E_FL), *fd = fl + 288, *misc = (uint32_t *)(lds + E_MISC);
    uint32_t *lltab = (uint32_t *)(lds + E_LLTAB), *dtab = (uint32_t *)(lds + E_DTAB);
    uint32_t *stage = (uint32_t *)(lds + E_STAGE);
    uint16_t *rle = (uint16_t *)(lds + E_HUF + H_RLE);
    uint32_t *clf = (uint32_t *)(lds + E_HUF + H_CLF);
    uint32_t *cltab = (uint32_t *)(lds + E_HUF + H_CLTAB);
    const uint32_t INF = 0xffffffffu;
    PH_DECL;
    PH_START();

    // ---- header fields from the code lengths in lltab / dtab (one whole wave): HLIT/HDIST, run-length
    // items, code-length code, and the header's size in bits
    auto header_fields = [&]() {
        const HufS TD = huf_scratch(lds + E_HUF + H_TD, 32);
        const uint32_t ulane = (uint32_t)lane;
        if (lane < 32) clf[lane] = 0;
        // HLIT / HDIST: trailing unused symbols are not sent (at least 257 / 1)
        uint32_t hlit, hdist;
        {
            const unsigned long long mh = __ballot(256 + lane < 286 && (lltab[256 + lane] >> 16) != 0);
            hlit = 256 + 64 - (uint32_t)__builtin_clzll(mh | 1ull);  // symbol 256 is always used
            if (hlit < 257) hlit = 257;
            const unsigned long long md = __ballot(lane < 30 && (dtab[lane & 31] >> 16) != 0);
            hdist = 64 - (uint32_t)__builtin_clzll(md | 1ull);
        }
        const uint32_t tot = hlit + hdist;
        // code lengths of the concatenated alphabets, 64 per chunk; run starts
        uint32_t Lv[5];
        unsigned long long St[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const uint32_t i = (uint32_t)c * 64 + ulane;
            Lv[c] = i < hlit ? lltab[i] >> 16 : (i < tot ? dtab[(i - hlit) & 31] >> 16 : 0xffu);
        }
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const uint32_t i = (uint32_t)c * 64 + ulane;
            const uint32_t pl = c ? rl(Lv[c ? c - 1 : 0], 63) : 0xffffu;  // last of the previous chunk
            uint32_t pv = __shfl_up(Lv[c], 1, 64);
            if (lane == 0) pv = pl;
            St[c] = __ballot(i <= tot && (i == tot || Lv[c] != pv));  // index tot closes the last run
        }
        // items per run (greedy: 18 for 11..138 zeros, 17 for 3..10, 16 for 3..6 repeats)
        uint32_t run[5], cnt[5], off[5];
        uint32_t nr = 0;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const uint32_t i = (uint32_t)c * 64 + ulane;
            const bool start = i < tot && ((St[c] >> ulane) & 1ull);
            uint32_t nxt = tot;
            {
                const unsigned long long above = lane == 63 ? 0ull : (St[c] >> (ulane + 1)) << (ulane + 1);
                bool found = false;
                if (above) {
                    nxt = (uint32_t)c * 64 + (uint32_t)__builtin_ctzll(above);
                    found = true;
                }
#pragma unroll
                for (int c2 = c + 1; c2 < 5; ++c2)
                    if (!found && St[c2]) {
                        nxt = (uint32_t)c2 * 64 + (uint32_t)__builtin_ctzll(St[c2]);
                        found = true;
                    }
            }
            uint32_t r = start ? nxt - i : 0u, n = 0;
            if (start) {
                if (Lv[c] == 0) {
                    const uint32_t q = r / 138u, rem = r % 138u;
                    n = q + (rem >= 3 ? 1u : rem);
                } else {
                    const uint32_t q = (r - 1) / 6u, rem = (r - 1) % 6u;
                    n = 1 + q + (rem >= 3 ? 1u : rem);
                }
            }
            run[c] = r;
            cnt[c] = n;
            uint32_t inc = n;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = __shfl_up(inc, d, 64);
                if (lane >= d) inc += y;
            }
            off[c] = nr + inc - n;
            nr += rl(inc, 63);
        }
        // write the items, count the code-length symbols
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            if (cnt[c]) {
                uint32_t r = run[c], o = off[c];
                const uint32_t v = Lv[c];
                if (v == 0) {
                    uint32_t n18 = 0, n17 = 0, n0 = 0;
                    while (r >= 11) {
                        const uint32_t t = r > 138 ? 138 : r;
                        rle[o++] = (uint16_t)(18 | ((t - 11) << 8));
                        ++n18;
                        r -= t;
                    }
                    if (r >= 3) {
                        rle[o++] = (uint16_t)(17 | ((r - 3) << 8));
                        ++n17;
                        r = 0;
                    }
                    while (r--) {
                        rle[o++] = 0;
                        ++n0;
                    }
                    if (n18) atomicAdd(&clf[18], n18);
                    if (n17) atomicAdd(&clf[17], n17);
                    if (n0) atomicAdd(&clf[0], n0);
                } else {
                    uint32_t nv = 1, n16 = 0;
                    rle[o++] = (uint16_t)v;
                    --r;
                    while (r >= 3) {
                        const uint32_t t = r > 6 ? 6 : r;
                        rle[o++] = (uint16_t)(16 | ((t - 3) << 8));
                        ++n16;
                        r -= t;
                    }
                    while (r--) {
                        rle[o++] = (uint16_t)v;
                        ++nv;
                    }
                    atomicAdd(&clf[v], nv);
                    if (n16) atomicAdd(&clf[16], n16);
                }
            }
        }
        asm volatile("" ::: "memory");
        PH_END(5);
        // ---- code-length code (19 symbols, 7 bits) ----
        {
            uint32_t f = lane < 19 ? clf[lane] : 0u;
            const unsigned long long nz = __ballot(f != 0);
            if (__builtin_popcountll(nz) < 2 && ulane == ((nz & 1ull) ? 1u : 0u)) f = 1;
            uint32_t key = f ? ((f << 9) | ulane) : INF;
            key = wave_sort64(key);
            const uint32_t mC = (uint32_t)__builtin_popcountll(__ballot(key != INF));
            if (lane < 32) TD.sorted[lane] = key;
            asm volatile("" ::: "memory");
            wave_huffman<1>(TD, mC, 19, 7, cltab);
        }
        asm volatile("" ::: "memory");
        PH_END(6);
        // ---- size of the header: 17 bits, 3 per sent code-length length, the items
        uint32_t hclen;
        {
            const uint32_t cll = lane < 19 ? cltab[hclen_order(ulane)] >> 16 : 0u;
            const unsigned long long mu = __ballot(cll != 0);
            hclen = 64 - (uint32_t)__builtin_clzll(mu | 1ull);
            if (hclen < 4) hclen = 4;
        }
        uint32_t hbits = 0;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const uint32_t t = (uint32_t)c * 64 + ulane;
            if (t < nr) {
                const uint32_t sym = rle[t] & 0xffu;
                hbits += (cltab[sym] >> 16) + (sym == 16 ? 2u : sym == 17 ? 3u : sym == 18 ? 7u : 0u);
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) hbits += __shfl_xor(hbits, d, 64);
        if (lane == 0) {
            misc[M_META] = hlit;
            misc[M_META + 1] = hdist;
            misc[M_META + 2] = hclen;
            misc[M_META + 3] = nr;
            misc[M_META + 4] = 17 + 3 * hclen + hbits;
        }
    };

    // ---- code tables of block gb into lltab / dtab, header fields into rle / cltab / misc[M_META..];
    // returns the block's size in bits (header + tokens)
    auto build_block = [&](uint32_t gb, uint32_t ntok) -> uint32_t {
        for (int i = tid; i < 320; i += EWG) fl[i] = P.hist[(size_t)gb * 320 + i];
        group_sync<NT>();
        PH_END(1);
        if (P.ctype == 2) {
            const HufS TL = huf_scratch(lds + E_HUF + H_TL, 288), TD = huf_scratch(lds + E_HUF + H_TD, 32);
            // every code set gets at least two symbols, so it is complete: a block that
            // holds only the end-of-block symbol gains literal 0, and the distance
            // alphabet is topped up to two used symbols
            if (tid == 0 && ntok < 2) fl[0] = 1;
            if (wave == (NT == 64 ? 0 : 1)) {
                const unsigned long long nz = __ballot(lane < 30 && fd[lane] != 0);
                const int c = __builtin_popcountll(nz);
                if (c < 2 && lane == 0) {
                    if (c == 0) fd[0] = 1;
                    fd[(c == 0 || (nz & 1ull)) ? 1 : 0] = 1;
                }
            }
            group_sync<NT>();
            // ---- sort keys (weight << 9 | symbol, unused = INF) ----
            for (int i = tid; i < 288; i += EWG)
                TL.niw[i] = (i < 286 && fl[i]) ? ((fl[i] << 9) | (uint32_t)i) : INF;
            if (tid < 32) TD.niw[tid] = (tid < 30 && fd[tid]) ? ((fd[tid] << 9) | (uint32_t)tid) : INF;
            group_sync<NT>();
            if constexpr (NT == 256) {
                // ---- rank sort: a key's rank is the number of smaller keys ----
                uint32_t rk0 = 0, rk1 = 0, key0, key1 = INF;
                {
                    key0 = TL.niw[tid];
                    const bool second = tid < 32 || (tid >= 64 && tid < 96);  // lit/len 256.., dist keys
                    const uint32_t *k1src = tid < 32 ? TL.niw + 256 + tid : TD.niw + (tid - 64);
                    if (second) key1 = *k1src;
                    const uint4 *kp = (const uint4 *)TL.niw;
#pragma unroll 4
                    for (int j = 0; j < 72; ++j) {
                        const uint4 q = kp[j];
                        rk0 += (q.x < key0) + (q.y < key0) + (q.z < key0) + (q.w < key0);
                    }
                    if (tid < 32) {
                        for (int j = 0; j < 72; ++j) {
                            const uint4 q = kp[j];
                            rk1 += (q.x < key1) + (q.y < key1) + (q.z < key1) + (q.w < key1);
                        }
                    } else if (second) {
                        for (int j = 0; j < 32; ++j) rk1 += TD.niw[j] < key1;
                    }
                }
                // a padding key's rank is the number of used symbols
                if (tid == 31) misc[M_NTOK] = rk1;  // lit/len key 287
                if (tid == 95) misc[M_TMP] = rk1;   // dist key 31
                if (key0 != INF) TL.sorted[rk0] = key0;
                if (key1 != INF) (tid < 32 ? TL.sorted : TD.sorted)[rk1] = key1;
            } else {
                // one wave: the used symbols are packed first (a small unit uses few), then sorted
                // in registers when they fit a wave, else ranked against each other
                const unsigned long long lt = (1ull << lane) - 1ull;
                uint32_t *U = TL.rank;  // free until the tree is built
                uint32_t m = 0;
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const uint32_t i = (uint32_t)c * 64 + (uint32_t)lane;
                    const uint32_t key = i < 288 ? TL.niw[i] : INF;
                    const unsigned long long mk = __ballot(key != INF);
                    if (key != INF) U[m + (uint32_t)__builtin_popcountll(mk & lt)] = key;
                    m += (uint32_t)__builtin_popcountll(mk);
                }
                group_sync<NT>();
                if (m <= 64) {
                    uint32_t key = (uint32_t)lane < m ? U[lane] : INF;
                    key = wave_sort64(key);
                    if ((uint32_t)lane < m) TL.sorted[lane] = key;
                } else {
                    const uint32_t m4 = (m + 3) & ~3u;
                    if ((uint32_t)lane < m4 - m) U[m + lane] = INF;  // pad to whole uint4 reads
                    group_sync<NT>();
                    const uint4 *kp = (const uint4 *)U;
                    for (uint32_t i = (uint32_t)lane; i < m; i += 64) {
                        const uint32_t key = U[i];
                        uint32_t rk = 0;
                        for (uint32_t j = 0; j < m4 / 4; ++j) {
                            const uint4 q = kp[j];
                            rk += (q.x < key) + (q.y < key) + (q.z < key) + (q.w < key);
                        }
                        TL.sorted[rk] = key;
                    }
                }
                uint32_t dk = lane < 32 ? TD.niw[lane] : INF;
                dk = wave_sort64(dk);
                if (lane < 32) TD.sorted[lane] = dk;
                const uint32_t md = (uint32_t)__builtin_popcountll(__ballot(dk != INF));  // all lanes vote
                if (lane == 0) {
                    misc[M_NTOK] = m;
                    misc[M_TMP] = md;
                }
            }
            group_sync<NT>();
            PH_END(2);
            const uint32_t mL = misc[M_NTOK], mD = misc[M_TMP];
            // ---- the two trees, one wave each ----
            // the tree builders and the header are one or two waves' serial work that the rest of the
            // workgroup waits for at the barrier: they go first among the waves the other workgroups of
            // the CU have on the same SIMDs (emit kernel 5.14 -> 4.60 ms on 64 KiB units)
            __builtin_amdgcn_s_setprio(3);
            if (wave == 0) wave_huffman<5>(TL, mL, 286, 15, lltab);
            if (NT == 64) PH_END(3);
            if (wave == (NT == 64 ? 0 : 1)) wave_huffman<1>(TD, mD, 30, 15, dtab);
            group_sync<NT>();
            PH_END(4);
            if (wave == 0) header_fields();
            __builtin_amdgcn_s_setprio(0);
        } else {
            // fixed Huffman codes (RFC 1951 3.2.6; FixedHuffmanTable rawdeflate.js:48-68)
            for (int sy = tid; sy < 288; sy += EWG) {
                uint32_t c, l;
                if (sy <= 143) {
                    c = 0x30 + sy;
                    l = 8;
                } else if (sy <= 255) {
                    c = 0x190 + sy - 144;
                    l = 9;
                } else if (sy <= 279) {
                    c = sy - 256;
                    l = 7;
                } else {
                    c = 0xc0 + sy - 280;
                    l = 8;
                }
                lltab[sy] = (__brev(c) >> (32 - l)) | (l << 16);
            }
            if (tid < 32) dtab[tid] = (__brev((uint32_t)tid) >> 27) | (5u << 16);
            if (tid == 0) misc[M_META + 4] = 3;
        }
        group_sync<NT>();
        if constexpr (!DIRECT) return 0u;
        // ---- bits of the tokens: counts (as the search kernel left them: the top-ups above are
        // codes without tokens) times code length plus extra bits
        uint32_t mine = 0;
        for (int i = tid; i < 320; i += EWG) {
            const uint32_t n = P.hist[(size_t)gb * 320 + i];
            if (!n) continue;
            uint32_t bits;
            if (i < 288) {
                bits = lltab[i] >> 16;
                if (i >= 265 && i < 285) bits += ((uint32_t)i - 261u) >> 2;
            } else {
                const uint32_t d = (uint32_t)i - 288u;
                bits = (dtab[d] >> 16) + (d >= 4 ? (d >> 1) - 1u : 0u);
            }
            mine += n * bits;
        }
        uint32_t total;
        (void)group_scan_excl<NT>(mine, &total, misc + M_SCAN);
        group_sync<NT>();
        return total + misc[M_META + 4];
    };

    for (;;) {
        if (tid == 0) {
            // a range of units, or the entries of a list (a round emitted per size class)
            const uint32_t ticket = atomicAdd(P.counter, 1u);
            misc[M_UNIT] = P.unit_list ? (ticket < P.list_len ? P.unit_list[ticket] : 0xffffffffu) : P.unit_first + ticket;
        }
        group_sync<NT>();
        const uint32_t u = misc[M_UNIT];
        group_sync<NT>();
        if (u >= P.unit_end) break;
        PH_END(0);
        const uint32_t n = (uint32_t)(P.in_off[u + 1] - P.in_off[u]);
        const bool final_unit = P.unit_final[u] != 0;
        const uint32_t nblk = n ? (n + ZGPU_BLOCK_BYTES - 1) / ZGPU_BLOCK_BYTES : 1u;
        const uint32_t gb0 = P.blk_first[u];
        const uint16_t *tok0 = P.tok + P.tok_off[u];
        uint32_t unit_bytes = 0;
        uint32_t cap_words = 0xffffffffu;  // slots: room of the unit's slot
        if constexpr (DIRECT) {
            // ---- (1) size of the unit ----
            uint32_t unit_bits = 0;
            for (uint32_t blk = 0; blk < nblk; ++blk) {
                const uint32_t gb = gb0 + blk;
                unit_bits += build_block(gb, P.ntok[gb]);
                if (nblk > 1 && P.ctype == 2) {
                    // park the code tables where the block's histogram was (320 words either way); the
                    // header fields are rebuilt from them when the block is emitted
                    for (int i = tid; i < 320; i += EWG) P.hist[(size_t)gb * 320 + i] = lltab[i];  // lltab, dtab adjoin
                    group_sync<NT>();
                }
            }
            // byte-align; a chunk that is followed by more of the same stream ends with an empty stored block
            unit_bytes = final_unit ? (unit_bits + 7) >> 3 : ((unit_bits + 3 + 7) >> 3) + 4;
            // ---- (2) its place ----
            if (wave == 0) {
                const uint64_t e = lookback_offset(P.state, u, unit_bytes);
                if (lane == 0) {
                    misc[M_EXCL] = (uint32_t)e;
                    misc[M_EXCL + 1] = (uint32_t)(e >> 32);
                    P.out_off[u] = e;
                    if (u + 1 == P.n_total) P.out_off[u + 1] = e + unit_bytes;
                    P.out_size[u] = unit_bytes;
                }
            }
        } else {
            const uint64_t so = P.slot_off[u];
            cap_words = (uint32_t)((P.slot_off[u + 1] - so) >> 2);
            if (tid == 0) {
                misc[M_EXCL] = (uint32_t)so;
                misc[M_EXCL + 1] = (uint32_t)(so >> 32);
                misc[M_OVF] = 0;
            }
        }
        for (int i = tid; i < STAGE_WORDS; i += EWG) stage[i] = 0;
        group_sync<NT>();
        const uint64_t excl = (uint64_t)misc[M_EXCL] | ((uint64_t)misc[M_EXCL + 1] << 32);
        const bool placed = excl != ~0ull, fits = placed && (!DIRECT || excl + unit_bytes <= P.out_cap);
        const uint32_t head = (uint32_t)(excl & 3u);  // bytes of the first word that belong to the unit before
        uint32_t *out32 = (uint32_t *)(P.out + (excl - head));
        uint32_t bitpos = 8 * head;  // bits so far, counted from the first word's start (uniform)
        uint32_t base_word = 0;      // output word index of stage[0] (uniform)
        // a full word goes out whole unless it is the unit's first and shared with the unit before
        auto store_word = [&](uint32_t wi, uint32_t w) {
            if constexpr (!DIRECT) {
                if (wi < cap_words)
                    out32[wi] = w;
                else
                    misc[M_OVF] = 1;
            } else if (wi == 0 && head) {
                uint8_t *b = (uint8_t *)out32;
                for (uint32_t k = head; k < 4; ++k) b[k] = (uint8_t)(w >> (8 * k));
            } else
                out32[wi] = w;
        };
        // ---- (3) header and tokens, block by block ----
        for (uint32_t blk = 0; blk < nblk; ++blk) {
            const uint32_t gb = gb0 + blk;
            const uint16_t *tokbuf = tok0 + (uint64_t)blk * ZGPU_BLOCK_TOK;
            const uint32_t ntok = P.ntok[gb];
            const uint32_t bfinal = (blk + 1 == nblk && final_unit) ? 1u : 0u;
            if constexpr (!DIRECT) {
                (void)build_block(gb, ntok);
            } else if (nblk > 1 && P.ctype == 2) {
                // the tables parked at sizing time come back; the header fields are rebuilt from them
                for (int i = tid; i < 320; i += EWG) lltab[i] = P.hist[(size_t)gb * 320 + i];
                group_sync<NT>();
                if (wave == 0) header_fields();
                group_sync<NT>();
            }
            // stage[] holds the unit's partial last word at [0] and is zero elsewhere
            const uint32_t hb = bitpos - base_word * 32;  // bit offset of the header in stage[]
            if (wave == 0) {
                auto put_bits = [&](uint32_t rel, uint32_t v, uint32_t nb) {
                    // nb <= 17: at most two words
                    const uint32_t w = rel >> 5, sh = rel & 31;
                    const uint64_t x = (uint64_t)v << sh;
                    atomicOr(&stage[w], (uint32_t)x);
                    if (sh + nb > 32) atomicOr(&stage[w + 1], (uint32_t)(x >> 32));
                };
                const uint32_t ulane = (uint32_t)lane;
                if (P.ctype == 2) {
                    const uint32_t hlit = misc[M_META], hdist = misc[M_META + 1], hclen = misc[M_META + 2],
                                   nr = misc[M_META + 3];
                    const uint32_t cll = lane < 19 ? cltab[hclen_order(ulane)] >> 16 : 0u;
                    if (lane == 0)
                        put_bits(hb, bfinal | (2u << 1) | ((hlit - 257) << 3) | ((hdist - 1) << 8) | ((hclen - 4) << 13), 17);
                    if (ulane < hclen) put_bits(hb + 17 + 3 * ulane, cll, 3);
                    uint32_t pos = hb + 17 + 3 * hclen;
#pragma unroll
                    for (int c = 0; c < 5; ++c) {
                        const uint32_t t = (uint32_t)c * 64 + ulane;
                        uint32_t v = 0, nb = 0;
                        if (t < nr) {
                            const uint32_t it = rle[t], sym = it & 0xffu, ex = it >> 8;
                            const uint32_t e = cltab[sym];
                            v = e & 0xffffu;
                            nb = e >> 16;
                            v |= ex << nb;
                            nb += sym == 16 ? 2u : sym == 17 ? 3u : sym == 18 ? 7u : 0u;
                        }
                        uint32_t inc = nb;
#pragma unroll
                        for (int d = 1; d < 64; d <<= 1) {
                            const uint32_t y = __shfl_up(inc, d, 64);
                            if (lane >= d) inc += y;
                        }
                        if (nb) put_bits(pos + inc - nb, v, nb);
                        pos += rl(inc, 63);
                    }
                } else if (lane == 0)
                    put_bits(hb, bfinal | (1u << 1), 3);
            }
            bitpos += misc[M_META + 4];
            group_sync<NT>();
            PH_END(7);
            // ---- emit tokens: four slots per thread and pass ----
            // bits of a pass are OR-merged into the staging words at their prefix-sum
            // offsets; full words leave with coalesced stores and are cleared by the
            // thread that stored them, the partial last word moves to the front.
            // A slot is a literal / end-of-block, the first half of a match (bit 15) or its
            // second half (the slot before it has bit 15; a distance - 1 never has).
            auto lit_bits = [&](uint32_t v, uint64_t &bits, uint32_t &nb) {
                const uint32_t e = lltab[v];
                bits = e & 0xffff;
                nb = e >> 16;
            };
            auto match_bits = [&](uint32_t first, uint32_t second, uint64_t &bits, uint32_t &nb) {
                uint32_t sym, eb, ev;
                len_sym(first & 0xff, sym, eb, ev);
                uint32_t e = lltab[sym];
                bits = e & 0xffff;
                nb = e >> 16;
                bits |= (uint64_t)ev << nb;
                nb += eb;
                dist_sym(second, sym, eb, ev);
                e = dtab[sym];
                bits |= (uint64_t)(e & 0xffff) << nb;
                nb += e >> 16;
                bits |= (uint64_t)ev << nb;
                nb += eb;
            };
            auto stage_or = [&](uint32_t rel, uint64_t bits, uint32_t nb) {
                if (!nb) return;
                const uint32_t w = rel >> 5, sh = rel & 31;
                const uint64_t lo = bits << sh;
                atomicOr(&stage[w], (uint32_t)lo);
                if ((uint32_t)(lo >> 32)) atomicOr(&stage[w + 1], (uint32_t)(lo >> 32));
                if (sh && (nb + sh > 64)) atomicOr(&stage[w + 2], (uint32_t)(bits >> (64 - sh)));
            };
            for (uint32_t t0 = 0; t0 < ntok; t0 += ETOK) {
                uint64_t sb[4] = {0, 0, 0, 0};
                uint32_t sn[4] = {0, 0, 0, 0};
                const uint32_t ti = t0 + 4 * (uint32_t)tid;
                if (ti < ntok) {
                    const uint2 q = *(const uint2 *)(tokbuf + ti);  // ti is a multiple of 4, as are block regions
                    const uint32_t v[4] = {q.x & 0xffffu, q.x >> 16, q.y & 0xffffu, q.y >> 16};
                    uint32_t before = ti ? tokbuf[ti - 1] : 0u;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        if (ti + s4 < ntok && !(before & 0x8000u)) {
                            if (v[s4] & 0x8000u)
                                match_bits(v[s4], s4 < 3 ? v[(s4 + 1) & 3] : (uint32_t)tokbuf[ti + 4], sb[s4], sn[s4]);
                            else
                                lit_bits(v[s4], sb[s4], sn[s4]);
                        }
                        before = v[s4];
                    }
                }
                uint32_t total;
                const uint32_t mine = sn[0] + sn[1] + sn[2] + sn[3];
                const uint32_t off = group_scan_excl<NT>(mine, &total, misc + M_SCAN);
                uint32_t rel = bitpos + off - base_word * 32;
                // neighbours that fit one 64-bit piece together go as one
                uint64_t acc = 0;
                uint32_t accn = 0;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    if (!sn[s4]) continue;
                    if (accn + sn[s4] > 64) {
                        stage_or(rel, acc, accn);
                        rel += accn;
                        acc = 0;
                        accn = 0;
                    }
                    acc |= sb[s4] << accn;
                    accn += sn[s4];
                }
                stage_or(rel, acc, accn);
                group_sync<NT>();
                bitpos += total;
                const uint32_t full = (bitpos >> 5) - base_word;
                for (uint32_t i = tid; i < full; i += EWG) {
                    if (fits) store_word(base_word + i, stage[i]);
                    stage[i] = 0;
                }
                if (tid == 0 && full) {
                    stage[0] = stage[full];
                    stage[full] = 0;
                }
                base_word += full;
                // the two barriers of the next pass's prefix sum separate this from its ORs
            }
            group_sync<NT>();
            PH_END(8);
        }
        // ---- finish the unit: byte-align, or an empty stored block for a chunk that is
        // followed by more of the same stream ----
        if (tid == 0) {
            uint32_t bp = bitpos, cw = stage[0], bwd = base_word;
            stage[0] = 0;
            auto put = [&](uint32_t v, uint32_t nb) {
                // serial writer straight to the output (at most a few words)
                while (nb) {
                    uint32_t sh = bp & 31, take = 32 - sh < nb ? 32 - sh : nb;
                    cw |= (v & (take == 32 ? ~0u : ((1u << take) - 1))) << sh;
                    v = take == 32 ? 0 : v >> take;
                    nb -= take;
                    bp += take;
                    if ((bp & 31) == 0) {
                        if (fits) store_word(bwd, cw);
                        ++bwd;
                        cw = 0;
                    }
                }
            };
            if (!final_unit) {
                put(0, 3);  // BFINAL=0, BTYPE=00
                put(0, (8 - (bp & 7)) & 7);
                put(0x0000, 16);
                put(0xffff, 16);
            } else
                put(0, (8 - (bp & 7)) & 7);
            if constexpr (DIRECT) {
                if ((bp & 31) && fits) {
                    // the last word is shared with the unit behind: bytewise, like a shared first word
                    uint8_t *b = (uint8_t *)(out32 + bwd);
                    for (uint32_t k = (bwd == 0 ? head : 0u); k < ((bp & 31) >> 3); ++k) b[k] = (uint8_t)(cw >> (8 * k));
                }
                P.status[u] = !placed ? 1u : !fits ? 2u : ((bp >> 3) - head != unit_bytes ? 1u : 0u);
            } else {
                if (bp & 31) store_word(bwd, cw);
                P.out_size[u] = bp >> 3;
                P.status[u] = misc[M_OVF] ? 1u : 0u;
            }
        }
        group_sync<NT>();
        PH_END(9);
    }
    PH_FLUSH();
}

}  // namespace

extern "C" __global__ void __launch_bounds__(256, ZGPU_EMIT_WAVES)
zgpu_emit_kernel(EParams P) {
    __shared__ __align__(16) uint8_t lds[emit_lds_bytes(256)];
    emit_units<256, true>(P, lds);
}
// one wave per unit, four units per workgroup: chosen for rounds of small units
extern "C" __global__ void __launch_bounds__(256, 4)
zgpu_emit_small_kernel(EParams P) {
    __shared__ __align__(16) uint8_t lds[4 * emit_lds_bytes(64)];
    emit_units<64, false>(P, lds + (threadIdx.x >> 6) * emit_lds_bytes(64));
}
// workgroup per unit into slots: plans that mix rounds of small units with rounds of larger ones
extern "C" __global__ void __launch_bounds__(256, 8)
zgpu_emit_slots_kernel(EParams P) {
    __shared__ __align__(16) uint8_t lds[emit_lds_bytes(256)];
    emit_units<256, false>(P, lds);
}

// ---- content class of a unit: "dense" = few byte values carry the unit, or it is made of short runs ----
// In such data every 4-byte key has hundreds to thousands of earlier occurrences and the longest of them
// is a few bytes longer than the longest of the nearest eight, at *every* position: the bounded walk of
// the search kernel leaves 3-47 % of size against the reference's exhaustive one (src/rawdeflate.js:370-395),
// where on other data it leaves nothing.  Units of that class are searched by the thorough kernel variant
// over the list written here, with the settings of their class (the byte written to `taken`), and passed
// over by the default search.
// One wave per unit looks at ZGPU_CLS_CHUNKS chunks of 64 bytes spread evenly over the unit:
//   * the chance that two of the sampled bytes are equal (sum of squared counts) is at least 0.155 --
//     what six equally likely values give; seven give 0.143, text 0.05-0.08, which the default search
//     serves within the bar -- or at least half of the sampled bytes equal their successor (runs of any
//     number of values);
//   * and fewer than 60 % of the chunks are one byte value throughout: runs of hundreds of bytes reach
//     the maximum match length at the first candidates of their chains and need no second search, which
//     takes six times as long on them (runs of up to 512 bytes of four values, a tenth of the benchmark's
//     units: the share is 0.77 +- 0.07 and never below 0.56 in 800 units; up to 128 bytes of two values,
//     several hundred where equal neighbours merge: 0.5 +- 0.09, and 1.09-1.12 x the reference's size
//     without the second search).
// How many distinct byte values the sample holds decides how much the unit gets: up to eight, it is the kind
// of data the pass is for; with more -- zeros and small numbers in the tables of a binary, spaces in indented
// text -- half of the units of a shared library qualify, and they get a short second search only.
// The first two are taken over the chunks that are not one value throughout: a unit that is long runs in
// part and text for the rest has few values by count, and is not dense (the second search is at its slowest
// on long runs: a batch of such units took 300 ms instead of 24 when they were counted in).
#define ZGPU_CLS_CHUNKS 32
#define ZGPU_CLS_MAX_VALUES 8u
// (4096 for most of the round: two-valued files of 300-4000 bytes then came out at 1.07-1.20 x the reference's
// size under FIXED; classified, at 1.005-1.016 x.  The samples of a unit shorter than 2 KiB overlap.)
#define ZGPU_CLS_MIN_BYTES 256u
// units of the third class with more than this many byte values in the sample are keyed on three bytes (a skewed
// alphabet of eight or sixteen values is not: its three-byte chains are all short matches, and the sixty-four
// nearest of them hide the long ones -- 1.02 -> 1.05 x the reference's size)
#ifndef ZGPU_DENSE_KEY3_VALUES
#define ZGPU_DENSE_KEY3_VALUES 16u
#endif
#ifndef ZGPU_DENSE_KEY3_BELOW
#define ZGPU_DENSE_KEY3_BELOW 3.5f
#endif
#ifndef ZGPU_CLS_TABLE_MAX_BYTES
#define ZGPU_CLS_TABLE_MAX_BYTES 16384u
#endif
#ifndef ZGPU_DENSE_LONG_KEY
#define ZGPU_DENSE_LONG_KEY 1
#endif
#ifndef ZGPU_CLS_ONE_VALUE_PCT
#define ZGPU_CLS_ONE_VALUE_PCT 60u
#endif
#define ZGPU_CLS_SAMPLES (64u * ZGPU_CLS_CHUNKS)
extern "C" __global__ void __launch_bounds__(256)
zgpu_classify_kernel(const uint8_t *in, const uint64_t *in_off, uint32_t unit_first, uint32_t unit_end,
                     uint32_t *dense_list, uint32_t *dense_count, uint8_t *taken, uint32_t fixed_codes) {
    __shared__ uint32_t count[4][256];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t u = unit_first + blockIdx.x * 4u + w;
    for (int i = 0; i < 4; ++i) count[w][i * 64 + lane] = 0;
    __syncthreads();
    const uint64_t uo = u < unit_end ? in_off[u] : 0;
    const uint64_t un = u < unit_end ? in_off[u + 1] - uo : 0;
    const bool live = un >= ZGPU_CLS_MIN_BYTES;  // uniform per wave; a unit has less than 4 GiB (ZGPU_MAX_UNIT_BYTES)
    uint32_t one_value = 0, equal = 0;
    if (live) {
        const uint32_t n = (uint32_t)un, stride = (n - 65u) / (ZGPU_CLS_CHUNKS - 1);
        const uint8_t *src = in + uo;
        // all samples are requested before the first is looked at (the loop below would otherwise pay one
        // memory round trip per chunk: 0.28 -> 0.18 ms per 32768 small files); lane l reads byte l of the chunk,
        // lane 63 also the byte behind it
        uint8_t b[ZGPU_CLS_CHUNKS], last[ZGPU_CLS_CHUNKS];
#pragma unroll
        for (uint32_t c = 0; c < ZGPU_CLS_CHUNKS; ++c) {
            const uint32_t at = c * stride + lane;  // at + 1 <= (chunks - 1) * stride + 64 <= n - 1
            b[c] = src[at];
            last[c] = lane == 63u ? src[at + 1] : (uint8_t)0;
        }
#pragma unroll
        for (uint32_t c = 0; c < ZGPU_CLS_CHUNKS; ++c) {
            const uint32_t b0 = b[c];
            const uint32_t nb = (uint32_t)__shfl_down((int)b0, 1);
            const uint32_t b1 = lane == 63u ? (uint32_t)last[c] : nb;
            const unsigned long long same = __ballot(b0 == b1);
            if (same == ~0ull) {
                ++one_value;
            } else {  // (uniform) the statistics are those of the chunks that are not one value throughout
                atomicAdd(&count[w][b0], 1u);
                equal += (uint32_t)__builtin_popcountll(same);
            }
        }
    }
    __syncthreads();
    uint32_t squares = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t c = count[w][i * 64 + lane];
        squares += c * c;
    }
    for (int d = 32; d; d >>= 1) squares += __shfl_xor(squares, d);
    uint32_t values = 0;
    for (int i = 0; i < 4; ++i) values += (uint32_t)__builtin_popcountll(__ballot(count[w][i * 64 + lane] != 0u));
    const uint32_t used = 64u * (ZGPU_CLS_CHUNKS - one_value);  // samples behind the statistics
    // 0.155; a unit of more than eight values must reach 0.25 (indented source text has 0.15-0.17 for its
    // spaces and gains nothing; the tables of a binary have 0.3-0.7 for their zeros)
    const bool few_values = values <= ZGPU_CLS_MAX_VALUES ? squares * 200ull >= 31ull * used * used
                                                          : squares * 4ull >= (unsigned long long)used * used;
    const bool short_runs = equal * 2u >= used;
    if (live && lane == 0 && (few_values || short_runs) && one_value * 100u < ZGPU_CLS_ONE_VALUE_PCT * ZGPU_CLS_CHUNKS) {
        // three classes (see the launch): at most six byte values; seven or eight in short runs; everything else
        const uint32_t which = values <= 6u ? 0u : values <= ZGPU_CLS_MAX_VALUES && short_runs ? 1u : 2u;
        dense_list[atomicAdd(dense_count, 1u)] = u;
        // Units of at most six values: how many bytes key their chains.  With q the chance that two bytes are equal, a
        // key of L bytes leaves n * q^L candidates per chain on average; the longest key that still leaves 32 of them
        // (so that a match of at least that length is there to be found), from four bytes up to eight.
        uint32_t key_code = 0;
        if (which == 0u && ZGPU_DENSE_LONG_KEY) {
            const float q = (float)squares / ((float)used * (float)used);
            const float window = un < 32768ull ? (float)un : 32768.0f;
            const float len = __logf(window * (1.0f / 32.0f)) / -__logf(q < 0.999f ? q : 0.999f);
            key_code = len >= 8.0f ? 3u : len >= 6.0f ? 2u : len >= 5.0f ? 1u : 0u;
            // ... and three where even four leave too few -- small files, or five and six values in a few KiB: the
            // reference keys on three (src/rawdeflate.js:314-320) and a file of 300-3000 bytes lives on its three-byte
            // matches (five values in 1000 bytes: 1.06 x the reference's size under FIXED keyed on four, 1.000 x on
            // three).  Under FIXED only: with DYNAMIC codes the same files are 4-8 % *smaller* than the reference's
            // without those matches, and equal to it with them.
            if (fixed_codes && len < ZGPU_DENSE_KEY3_BELOW) key_code = 4u;  // (bit 6 of the byte)
        }
        // the first search passes the unit over; the second reads its class (and key length) here
        taken[u] = (uint8_t)(1u + which + (key_code << 4) + (which == 2u && values > ZGPU_DENSE_KEY3_VALUES ? 0x40u : 0u));
    } else if (ZGPU_FIRST_RUN_KEY && live && lane == 0 && equal * 4u >= used && one_value * 100u < 30u * ZGPU_CLS_CHUNKS) {
        taken[u] = 0x80u;  // not of the class, but a quarter of its bytes equal their successor (indented text)
    } else if (live && lane == 0 && !fixed_codes && un <= ZGPU_CLS_TABLE_MAX_BYTES && values > 16u &&
               count[w][0] * 10u >= used) {
        // Not of the class either, under DYNAMIC, small, of many byte values, and a tenth of the sample is the zero
        // byte: the tables of a binary (16-bit values in 32-bit slots), whose repeats are three bytes long.  Its first
        // search keys on three bytes, as all small units do under FIXED: slices of 3000-8000 bytes of shared libraries
        // were 1.014-1.022 x the reference's size in sum, three of sixteen above 1.03 x, and are 1.000-1.002 x with
        // none.  (Small text keyed that way comes out at the reference's size too -- but is 1-3 % smaller keyed on four.)
        taken[u] = 0x08u;
    }
}

// ---- stored blocks (CompressionType.NONE; makeNocompressBlock rawdeflate.js:131-153)
// The sizes follow from the unit sizes alone, so the host has the final offsets (slot_off) ready.
extern "C" __global__ void __launch_bounds__(256)
zgpu_stored_kernel(const uint8_t *in, const uint64_t *in_off, const uint64_t *slot_off,
                   uint8_t *out, uint64_t out_cap, uint64_t *out_off, uint32_t *out_size, uint32_t *status,
                   uint32_t n_units, uint32_t flags) {
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint8_t *src = in + in_off[u];
        const uint64_t n = in_off[u + 1] - in_off[u];
        const bool fits = slot_off[u + 1] <= out_cap;
        uint8_t *dst = out + slot_off[u];
        const bool final_unit = (flags & 1u) || ((flags & 4u) && u == n_units - 1) || !(flags & 2u);
        const uint64_t nblk = n ? (n + 65534) / 65535 : 1;
        for (uint64_t b = 0; fits && b < nblk; ++b) {
            const uint64_t o = b * 65535;
            const uint32_t len = (uint32_t)((n - o) < 65535 ? (n - o) : 65535);
            uint8_t *d = dst + b * 65540;
            if (threadIdx.x == 0) {
                d[0] = (b == nblk - 1 && final_unit) ? 1 : 0;
                d[1] = len & 0xff;
                d[2] = len >> 8;
                d[3] = ~len & 0xff;
                d[4] = (~len >> 8) & 0xff;
            }
            // payload: bytes up to a 4-byte boundary of the destination, then dwords put together from the
            // two source dwords that hold them (source and destination are aligned differently), then the tail
            uint8_t *d0 = d + 5;
            const uint8_t *s0 = src + o;
            uint32_t headn = (4u - (uint32_t)((uintptr_t)d0 & 3u)) & 3u;
            if (headn > len) headn = len;
            if (threadIdx.x < headn) d0[threadIdx.x] = s0[threadIdx.x];
            const uint32_t body = (len - headn) & ~3u;
            const uint8_t *sb = s0 + headn;
            uint32_t *dw = (uint32_t *)(d0 + headn);
            const uint32_t sa = (uint32_t)((uintptr_t)sb & 3u);
            const uint32_t *sw = (const uint32_t *)(sb - sa);
            for (uint32_t i = threadIdx.x; i < body / 4; i += blockDim.x)
                dw[i] = sa ? __builtin_amdgcn_alignbyte(sw[i + 1], sw[i], sa) : sw[i];
            for (uint32_t i = headn + body + threadIdx.x; i < len; i += blockDim.x) d0[i] = s0[i];
        }
        if (threadIdx.x == 0) {
            out_size[u] = (uint32_t)(n + 5 * nblk);
            out_off[u] = slot_off[u];
            if (u + 1 == n_units) out_off[u + 1] = slot_off[u + 1];
            status[u] = fits ? 0u : 2u;
        }
    }
}

// ---- exclusive scan of unit sizes -> contiguous offsets (single workgroup) --
extern "C" __global__ void __launch_bounds__(1024)
zgpu_scan_kernel(const uint32_t *sizes, uint64_t *out_off, uint32_t n) {
    __shared__ unsigned long long part[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        unsigned long long v = i < n ? sizes[i] : 0, x = v;
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) part[wave] = x;
        __syncthreads();
        unsigned long long w = 0, tot = 0;
        for (int k = 0; k < 16; ++k) {
            if (k < wave) w += part[k];
            tot += part[k];
        }
        const unsigned long long c = carry;
        if (i < n) out_off[i] = c + w + x - v;
        __syncthreads();
        if (threadIdx.x == 0) carry = c + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) out_off[n] = carry;
}

// ---- gather the bounded slots into one contiguous buffer -------------------
extern "C" __global__ void __launch_bounds__(256)
zgpu_compact_kernel(const uint8_t *out_tmp, const uint64_t *slot_off, const uint64_t *out_off,
                    uint8_t *out, uint64_t out_cap, uint32_t *status, uint32_t n_units) {
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint8_t *src = out_tmp + slot_off[u];
        const uint64_t o = out_off[u], len = out_off[u + 1] - o;
        if (o + len > out_cap) {
            if (threadIdx.x == 0) status[u] = 2;
            continue;
        }
        uint8_t *dst = out + o;
        // byte head to a 16-byte boundary of dst, then 16-byte stores fed by
        // unaligned-source dwords, then byte tail
        uint64_t headb = (16 - ((uintptr_t)dst & 15)) & 15;
        if (headb > len) headb = len;
        for (uint64_t i = threadIdx.x; i < headb; i += blockDim.x) dst[i] = src[i];
        const uint64_t body = (len - headb) & ~15ull;
        const uint8_t *s2 = src + headb;
        uint8_t *d2 = dst + headb;
        if ((((uintptr_t)s2) & 3) == 0) {
            for (uint64_t i = (uint64_t)threadIdx.x * 16; i < body; i += (uint64_t)blockDim.x * 16) {
                const uint32_t *sw = (const uint32_t *)(s2 + i);
                *(uint4 *)(d2 + i) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
            }
        } else {
            for (uint64_t i = (uint64_t)threadIdx.x * 16; i < body; i += (uint64_t)blockDim.x * 16) {
                const uintptr_t a = (uintptr_t)(s2 + i);
                const uint32_t *sw = (const uint32_t *)(a & ~(uintptr_t)3);
                const uint32_t sh = (uint32_t)(a & 3);
                uint32_t w0 = sw[0], w1 = sw[1], w2 = sw[2], w3 = sw[3], w4 = sw[4];
                *(uint4 *)(d2 + i) = make_uint4(__builtin_amdgcn_alignbyte(w1, w0, sh),
                                                __builtin_amdgcn_alignbyte(w2, w1, sh),
                                                __builtin_amdgcn_alignbyte(w3, w2, sh),
                                                __builtin_amdgcn_alignbyte(w4, w3, sh));
            }
        }
        for (uint64_t i = headb + body + threadIdx.x; i < len; i += blockDim.x) dst[i] = src[i];
    }
}

// ---- per-unit output offsets when long units were cut into pieces -------------
extern "C" __global__ void __launch_bounds__(256)
zgpu_unit_offsets_kernel(const uint64_t *piece_out_off, const uint32_t *first_piece, uint64_t *unit_out_off,
                         uint32_t n_units) {
    const uint32_t u = blockIdx.x * 256 + threadIdx.x;
    if (u <= n_units) unit_out_off[u] = piece_out_off[first_piece[u]];
}
