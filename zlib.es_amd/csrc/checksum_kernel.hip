// CRC-32 + Adler-32 in one pass over HBM (gfx950).
//
// Replaces CRC32.calc/update (src/crc32.js:14-42: byte-at-a-time table CRC,
// reflected polynomial 0xEDB88320, init/final xor 0xFFFFFFFF) and
// Adler32.mkHash/update (src/adler32.js:11-37) with bit-identical results.
//
// One persistent 256-thread workgroup folds one segment [off[i], off[i+1]) at
// a time.  Lanes read adjacent 16-byte pieces (fully coalesced); lane t owns
// pieces t, t+256, t+512, ... and keeps the CRC remainder of its pieces as
//     A <- A * x^(8*4096)  xor  crc16(piece)
// (4 lookups in the "advance 4096 bytes" tables + 16 lookups in the slice
// tables, all in LDS).  CRC is linear, so the 256 partial remainders are then
// folded pairwise with multiplications by x^(128*d) in GF(2)[x]/P.  The
// 0xFFFFFFFF preset is folded into the first four data bytes; leading bytes
// of the first (address-aligned) piece that lie before the segment are zeroed,
// which leaves a zero CRC state unchanged.  Adler-32 uses per-lane sums
// sum(b) and sum(q*b) (v_dot4_u32_u8) and s2 = n + n*sum(b) - sum(q*b).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "zgpu_dev.h"

namespace {
constexpr int T = ZGPU_CK_THREADS;
constexpr uint32_t POLY = 0xEDB88320u;
constexpr uint32_t ADLER_MOD = 65521u;

// (a * b) mod P in the reflected representation (bit 31 = x^0)
__device__ inline uint32_t gf_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
        p ^= (a & 0x80000000u) ? b : 0u;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? POLY : 0u);
    }
    return p;
}

__device__ inline uint32_t dot4(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot4(a, b, c, false);
}
}  // namespace

// tables: [0..16*256) slice tables Tk[b] = crc state of byte b followed by k
// zero bytes; [16*256..20*256) advance tables Sj[v] = (v << 8j) advanced by
// 16*256 bytes; [20*256..20*256+8) x^(128*d) for d = 1,2,4,..,128; then (wave variant)
// 4*256 advance tables for 16*64 bytes.
// NT threads share a segment: the workgroup (256), or one wave (64; four segments side by side
// in a workgroup, no barrier inside the loop: batches of small units, whose time is the chain of
// round trips and the fold rather than their bytes).
template <int NT>
__device__ __forceinline__ void checksum_segments(const uint8_t *in, const uint64_t *seg_off, uint32_t n_segs,
                                                  const uint32_t *tables, uint32_t *out_crc, uint32_t *out_adler) {
    __shared__ uint32_t tab[(NT == 64 ? 24 : 20) * 256];
    __shared__ uint32_t xpow[8];
    __shared__ uint32_t red_crc[T / 64];
    __shared__ unsigned long long red_sb[T / 64], red_sw[T / 64];
    __shared__ __align__(16) uint32_t tailw[T / 64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 20 * 256; i += T) tab[i] = tables[i];
    if (NT == 64)
        for (int i = tid; i < 4 * 256; i += T) tab[20 * 256 + i] = tables[20 * 256 + 8 + i];
    if (tid < 8) xpow[tid] = tables[20 * 256 + tid];
    __syncthreads();
    const uint32_t *S = tab + (NT == 64 ? 20 : 16) * 256;  // advance by 16 * NT bytes
    const int t = NT == 64 ? lane : tid;                     // index within the group that shares a segment
    const uint32_t first = NT == 64 ? blockIdx.x * (T / 64) + wave : blockIdx.x;
    const uint32_t stride = NT == 64 ? gridDim.x * (T / 64) : gridDim.x;

    for (uint32_t sg = first; sg < n_segs; sg += stride) {
        const uint64_t o = seg_off[sg], n = seg_off[sg + 1] - o;
        const uint8_t *p0 = in + o;
        uint32_t crc_state;
        unsigned long long sb = 0, sw = 0;  // sum b, sum q*b
        uint64_t tail_lo, tail_hi;          // byte range finished serially by the group's first thread
        uint32_t B = 0;
        if (n < 64) {
            tail_lo = 0;
            tail_hi = n;
            crc_state = 0xffffffffu;
        } else {
            const uint32_t a = (uint32_t)((uintptr_t)p0 & 15u);
            const uint8_t *vstart = p0 - a;
            const uint64_t kv = (a + n) >> 4;  // full 16-byte pieces (first may start before p0)
            const uint32_t pad = (uint32_t)((NT - (kv % NT)) % NT);
            const uint64_t rounds = (kv + pad) / NT;
            uint32_t A = 0;
            for (uint64_t r = 0; r < rounds; ++r) {
                const int64_t i = (int64_t)(r * NT + t) - (int64_t)pad;
                // advance by 16*NT bytes
                A = S[A & 0xff] ^ S[256 + ((A >> 8) & 0xff)] ^ S[512 + ((A >> 16) & 0xff)] ^
                    S[768 + (A >> 24)];
                if (i >= 0) {
                    uint4 v = *(const uint4 *)(vstart + 16 * (uint64_t)i);
                    uint32_t w[4] = {v.x, v.y, v.z, v.w};
                    const int64_t q0 = 16 * i - (int64_t)a;  // segment index of byte 0
                    if (i < 2) {
                        // zero bytes before the segment; fold the preset into bytes 0..3
                        uint32_t xr[4] = {0, 0, 0, 0};
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const int64_t q = q0 + k;
                            if (q < 0) w[k >> 2] &= ~(0xffu << (8 * (k & 3)));
                            if (q >= 0 && q < 4) xr[k >> 2] |= 0xffu << (8 * (k & 3));
                        }
                        // Adler sees the unmodified bytes
                        const uint32_t s = dot4(w[0], 0x01010101u, dot4(w[1], 0x01010101u,
                                           dot4(w[2], 0x01010101u, dot4(w[3], 0x01010101u, 0))));
                        const uint32_t j = dot4(w[0], 0x03020100u, dot4(w[1], 0x07060504u,
                                           dot4(w[2], 0x0b0a0908u, dot4(w[3], 0x0f0e0d0cu, 0))));
                        sb += s;
                        sw += (unsigned long long)((long long)q0 * (long long)s + (long long)j);
                        w[0] ^= xr[0];
                        w[1] ^= xr[1];
                        w[2] ^= xr[2];
                        w[3] ^= xr[3];
                    } else {
                        const uint32_t s = dot4(w[0], 0x01010101u, dot4(w[1], 0x01010101u,
                                           dot4(w[2], 0x01010101u, dot4(w[3], 0x01010101u, 0))));
                        const uint32_t j = dot4(w[0], 0x03020100u, dot4(w[1], 0x07060504u,
                                           dot4(w[2], 0x0b0a0908u, dot4(w[3], 0x0f0e0d0cu, 0))));
                        sb += s;
                        sw += (unsigned long long)q0 * s + j;
                    }
                    uint32_t c = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t x = w[k];
                        c ^= tab[(15 - 4 * k) * 256 + (x & 0xff)] ^
                             tab[(14 - 4 * k) * 256 + ((x >> 8) & 0xff)] ^
                             tab[(13 - 4 * k) * 256 + ((x >> 16) & 0xff)] ^
                             tab[(12 - 4 * k) * 256 + (x >> 24)];
                    }
                    A ^= c;
                }
            }
            // fold the lane remainders: lane t must be advanced by 16*(NT-1-t) bytes
#pragma unroll
            for (int d = 1, lv = 0; d < 64; d <<= 1, ++lv) {
                const uint32_t other = __shfl_down(A, d, 64);
                if ((lane & (2 * d - 1)) == 0) A = gf_mul(A, xpow[lv]) ^ other;
            }
            if constexpr (NT == 256) {
                if (lane == 0) red_crc[wave] = A;
                __syncthreads();
                if (tid == 0) {
                    uint32_t x = red_crc[0];
                    x = gf_mul(x, xpow[6]) ^ red_crc[1];
                    uint32_t y = red_crc[2];
                    y = gf_mul(y, xpow[6]) ^ red_crc[3];
                    B = gf_mul(x, xpow[7]) ^ y;
                }
            } else
                B = A;  // lane 0 holds the wave's remainder
            crc_state = B;
            tail_lo = kv * 16 - a;
            tail_hi = n;
        }
        // Adler partial sums across the group
        for (int d = 32; d > 0; d >>= 1) {
            sb += __shfl_down(sb, d, 64);
            sw += __shfl_down(sw, d, 64);
        }
        unsigned long long tsb = sb, tsw = sw;
        if constexpr (NT == 256) {
            if (lane == 0) {
                red_sb[wave] = sb;
                red_sw[wave] = sw;
            }
            __syncthreads();
            if (tid == 0) {
                tsb = 0;
                tsw = 0;
                for (int w = 0; w < T / 64; ++w) {
                    tsb += red_sb[w];
                    tsw += red_sw[w];
                }
            }
        }
        if (t == 0) {
            uint32_t c = crc_state;
            uint32_t *tw = tailw[NT == 64 ? wave : 0];
            // the bytes the lanes left over (fewer than 16, or a whole segment of fewer than 64): one
            // aligned 16-byte load per piece, then bytes out of LDS -- a byte load from global memory
            // per step would put a memory round trip into every step of this serial stretch
            for (uint64_t q = tail_lo; q < tail_hi;) {
                const uint32_t a2 = (uint32_t)((uintptr_t)(p0 + q) & 15u);
                *(uint4 *)tw = *(const uint4 *)(p0 + q - a2);
                const uint64_t left = tail_hi - q;
                const uint32_t cnt = 16u - a2 < left ? 16u - a2 : (uint32_t)left;
                for (uint32_t k = 0; k < cnt; ++k, ++q) {
                    const uint32_t b = ((const volatile uint8_t *)tw)[a2 + k];
                    c = (c >> 8) ^ tab[(c ^ b) & 0xff];
                    tsb += b;
                    tsw += q * b;
                }
            }
            out_crc[sg] = c ^ 0xffffffffu;
            const unsigned long long nm = n % ADLER_MOD, bm = tsb % ADLER_MOD, wm = tsw % ADLER_MOD;
            const uint32_t s1 = (uint32_t)((1 + bm) % ADLER_MOD);
            const uint32_t s2 = (uint32_t)((nm + nm * bm + (ADLER_MOD - wm)) % ADLER_MOD);
            out_adler[sg] = (s2 << 16) | s1;
        }
        if constexpr (NT == 256) __syncthreads();
    }
}

extern "C" __global__ void __launch_bounds__(ZGPU_CK_THREADS)
zgpu_checksum_kernel(const uint8_t *in, const uint64_t *seg_off, uint32_t n_segs,
                     const uint32_t *tables, uint32_t *out_crc, uint32_t *out_adler) {
    checksum_segments<256>(in, seg_off, n_segs, tables, out_crc, out_adler);
}
// one wave per segment (tables with the extra advance block only: see build_tables)
extern "C" __global__ void __launch_bounds__(ZGPU_CK_THREADS)
zgpu_checksum_wave_kernel(const uint8_t *in, const uint64_t *seg_off, uint32_t n_segs,
                          const uint32_t *tables, uint32_t *out_crc, uint32_t *out_adler) {
    checksum_segments<64>(in, seg_off, n_segs, tables, out_crc, out_adler);
}

// ---- wide variant for large buffers -------------------------------------------
// The kernel above is bound by its LDS lookups (20 per 16 bytes, ~7 LDS cycles
// each at random addresses).  This variant indexes its tables with 11 bits
// instead of 8: 12 lookups for the 128 bits of a piece + 3 to advance the 32-bit
// remainder = 15 -- and the 3 are paid once per ZGPU_CKW_PIECES consecutive pieces
// of a lane, whose remainders chain through their first four bytes: 13.5 lookups
// per 16 bytes with two.  The tables (15 x 2048 words, 120 KB) fill the LDS of a
// CU, so one 1024-thread workgroup runs per CU and every lane keeps the pieces of
// two more rounds in flight while it folds the current ones.
// tables: [0..12*2048) W_k[v] = remainder contributed by bits [11k, 11k+11) of a
// piece; [12*2048..15*2048) S_c[v] = (v << 11c) advanced from the end of a lane's
// last piece of one round to the end of its first piece of the next
// (16 * (PIECES * 1024 - PIECES + 1) bytes); then x^(128*2^lv) for lv = 0..ZGPU_CKW_XPOW-1;
// then the 256-entry byte table.
extern "C" __global__ void __launch_bounds__(ZGPU_CKW_THREADS)
zgpu_checksum_wide_kernel(const uint8_t *in, const uint64_t *seg_off, uint32_t n_segs,
                          const uint32_t *tables, uint32_t *out_crc, uint32_t *out_adler) {
    constexpr int TW = ZGPU_CKW_THREADS;
    constexpr int KP = ZGPU_CKW_PIECES, KP_LOG2 = KP == 1 ? 0 : KP == 2 ? 1 : 2;
    static_assert(KP == 1 || KP == 2 || KP == 4, "pieces per lane and round");
    __shared__ uint32_t tab[15 * 2048];
    __shared__ uint32_t xpow[ZGPU_CKW_XPOW];
    __shared__ uint32_t bytetab[256];
    __shared__ uint32_t red_crc[TW / 64];
    __shared__ unsigned long long red_sb[TW / 64], red_sw[TW / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 15 * 2048; i += TW) tab[i] = tables[i];
    if (tid < ZGPU_CKW_XPOW) xpow[tid] = tables[15 * 2048 + tid];
    if (tid < 256) bytetab[tid] = tables[15 * 2048 + ZGPU_CKW_XPOW + tid];
    __syncthreads();
    const uint32_t *S = tab + 12 * 2048;

    for (uint32_t sg = blockIdx.x; sg < n_segs; sg += gridDim.x) {
        const uint64_t o = seg_off[sg], n = seg_off[sg + 1] - o;
        const uint8_t *p0 = in + o;
        uint32_t crc_state = 0xffffffffu;
        unsigned long long sb = 0, sw = 0;  // sum b, sum q*b
        uint64_t tail_lo = 0, tail_hi = n;  // byte range finished serially by thread 0
        if (n >= 64) {
            const uint32_t a = (uint32_t)((uintptr_t)p0 & 15u);
            const uint8_t *vstart = p0 - a;
            const uint64_t kv = (a + n) >> 4;  // full 16-byte pieces (the first may start before p0)
            // A lane takes KP consecutive pieces per round: the remainder of one goes into the first four
            // bytes of the next (the usual sliced update), so only the first of them pays the three lookups
            // that carry the lane's remainder over the rest of the round: 3 + 12 KP lookups per KP pieces.
            constexpr uint32_t PER = (uint32_t)TW * KP;  // pieces per round
            const uint32_t pad = (uint32_t)((PER - (kv % PER)) % PER);
            const uint64_t rounds = (kv + pad) / PER;
            uint32_t A = 0;
            auto fetch = [&](uint64_t r, int k) -> uint4 {
                const int64_t i = (int64_t)((r * TW + tid) * KP + k) - (int64_t)pad;
                return i >= 0 ? *(const uint4 *)(vstart + 16 * (uint64_t)i) : make_uint4(0, 0, 0, 0);
            };
            // the pieces of two rounds per lane in flight beside the ones being folded
            uint4 cur[KP], nxt[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                cur[k] = fetch(0, k);
                nxt[k] = rounds > 1 ? fetch(1, k) : make_uint4(0, 0, 0, 0);
            }
            for (uint64_t r = 0; r < rounds; ++r) {
                uint4 nxt2[KP];
#pragma unroll
                for (int k = 0; k < KP; ++k) nxt2[k] = r + 2 < rounds ? fetch(r + 2, k) : make_uint4(0, 0, 0, 0);
                // the lane's remainder carried to the end of this round's first piece
                uint32_t R = S[A & 0x7ffu] ^ S[2048 + ((A >> 11) & 0x7ffu)] ^ S[4096 + (A >> 22)];
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    const int64_t i = (int64_t)((r * TW + tid) * KP + k) - (int64_t)pad;
                    uint32_t w[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
                    if (i >= 0) {
                        const int64_t q0 = 16 * i - (int64_t)a;  // segment index of byte 0
                        const uint32_t s = dot4(w[0], 0x01010101u, dot4(w[1], 0x01010101u,
                                           dot4(w[2], 0x01010101u, dot4(w[3], 0x01010101u, 0))));
                        const uint32_t j = dot4(w[0], 0x03020100u, dot4(w[1], 0x07060504u,
                                           dot4(w[2], 0x0b0a0908u, dot4(w[3], 0x0f0e0d0cu, 0))));
                        if (i < 2) {
                            // zero the bytes before the segment (Adler and CRC both skip them); fold
                            // the CRC preset into bytes 0..3 of the segment
                            uint32_t xr[4] = {0, 0, 0, 0};
#pragma unroll
                            for (int kk = 0; kk < 16; ++kk) {
                                const int64_t q = q0 + kk;
                                if (q < 0) w[kk >> 2] &= ~(0xffu << (8 * (kk & 3)));
                                if (q >= 0 && q < 4) xr[kk >> 2] |= 0xffu << (8 * (kk & 3));
                            }
                            const uint32_t s2 = dot4(w[0], 0x01010101u, dot4(w[1], 0x01010101u,
                                                dot4(w[2], 0x01010101u, dot4(w[3], 0x01010101u, 0))));
                            const uint32_t j2 = dot4(w[0], 0x03020100u, dot4(w[1], 0x07060504u,
                                                dot4(w[2], 0x0b0a0908u, dot4(w[3], 0x0f0e0d0cu, 0))));
                            sb += s2;
                            sw += (unsigned long long)((long long)q0 * (long long)s2 + (long long)j2);
                            w[0] ^= xr[0];
                            w[1] ^= xr[1];
                            w[2] ^= xr[2];
                            w[3] ^= xr[3];
                        } else {
                            sb += s;
                            sw += (unsigned long long)q0 * s + j;
                        }
                    }
                    // pieces before the segment are zeros and leave a zero remainder zero.  The first piece
                    // of the round adds its remainder to the carried one; the others take the remainder so
                    // far into their first four bytes.
                    if (k > 0) w[0] ^= R;
                    const uint32_t c2 = __builtin_amdgcn_alignbit(w[1], w[0], 22) & 0x7ffu;
                    const uint32_t c5 = __builtin_amdgcn_alignbit(w[2], w[1], 23) & 0x7ffu;
                    const uint32_t c8 = __builtin_amdgcn_alignbit(w[3], w[2], 24) & 0x7ffu;
                    const uint32_t W = tab[0 * 2048 + (w[0] & 0x7ffu)] ^ tab[1 * 2048 + ((w[0] >> 11) & 0x7ffu)] ^
                                       tab[2 * 2048 + c2] ^ tab[3 * 2048 + ((w[1] >> 1) & 0x7ffu)] ^
                                       tab[4 * 2048 + ((w[1] >> 12) & 0x7ffu)] ^ tab[5 * 2048 + c5] ^
                                       tab[6 * 2048 + ((w[2] >> 2) & 0x7ffu)] ^ tab[7 * 2048 + ((w[2] >> 13) & 0x7ffu)] ^
                                       tab[8 * 2048 + c8] ^ tab[9 * 2048 + ((w[3] >> 3) & 0x7ffu)] ^
                                       tab[10 * 2048 + ((w[3] >> 14) & 0x7ffu)] ^ tab[11 * 2048 + (w[3] >> 25)];
                    R = k > 0 ? W : R ^ W;
                }
                A = R;
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    cur[k] = nxt[k];
                    nxt[k] = nxt2[k];
                }
            }
            // fold the lane remainders: lane t must be advanced by 16*KP*(TW-1-t) bytes
#pragma unroll
            for (int d = 1, lv = KP_LOG2; d < 64; d <<= 1, ++lv) {
                const uint32_t other = __shfl_down(A, d, 64);
                if ((lane & (2 * d - 1)) == 0) A = gf_mul(A, xpow[lv]) ^ other;
            }
            if (lane == 0) red_crc[wave] = A;
            __syncthreads();
            if (wave == 0) {
                uint32_t B = lane < TW / 64 ? red_crc[lane] : 0u;
#pragma unroll
                for (int d = 1, lv = 6 + KP_LOG2; d < TW / 64; d <<= 1, ++lv) {
                    const uint32_t other = __shfl_down(B, d, 64);
                    if ((lane & (2 * d - 1)) == 0) B = gf_mul(B, xpow[lv]) ^ other;
                }
                if (lane == 0) crc_state = B;
            }
            tail_lo = kv * 16 - a;
        }
        // Adler partial sums across the workgroup
        for (int d = 32; d > 0; d >>= 1) {
            sb += __shfl_down(sb, d, 64);
            sw += __shfl_down(sw, d, 64);
        }
        if (lane == 0) {
            red_sb[wave] = sb;
            red_sw[wave] = sw;
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long tsb = 0, tsw = 0;
            for (int w = 0; w < TW / 64; ++w) {
                tsb += red_sb[w];
                tsw += red_sw[w];
            }
            uint32_t c = crc_state;
            for (uint64_t q = tail_lo; q < tail_hi; ++q) {
                const uint32_t b = p0[q];
                c = (c >> 8) ^ bytetab[(c ^ b) & 0xff];
                tsb += b;
                tsw += q * b;
            }
            out_crc[sg] = c ^ 0xffffffffu;
            const unsigned long long nm = n % ADLER_MOD, bm = tsb % ADLER_MOD, wm = tsw % ADLER_MOD;
            const uint32_t s1 = (uint32_t)((1 + bm) % ADLER_MOD);
            const uint32_t s2 = (uint32_t)((nm + nm * bm + (ADLER_MOD - wm)) % ADLER_MOD);
            out_adler[sg] = (s2 << 16) | s1;
        }
        __syncthreads();
    }
}
