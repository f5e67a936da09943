// DEFLATE encoder kernel for gfx950 (MI355X): one 1024-thread workgroup per
// unit (block / zip entry / stream chunk), persistent over a unit queue.
//
// Replaces, per unit, the reference's RawDeflate.compress() pipeline
// (src/rawdeflate.js:102-124): lz77 + searchLongestMatch_ (:283-395), the
// Huffman code construction (:474-605), the block header (:176-236) and the
// bit emission (:243-277, src/bitstream.js:42-95).
//
// Per unit the workgroup streams through the input in 1 KiB chunks:
//   load    input -> 64 KiB LDS ring (the 32 KiB window stays resident)
//   insert  hash every position, sort (hash,pos) pairs, link hash chains
//   search  one lane per position walks its chain (bounded depth), scoring
//           candidates with 4-byte xor/ctz compares against the LDS ring
//   parse   greedy / one-step-lazy successor per position, reachable set by
//           pointer doubling, tokens compacted with a workgroup prefix sum,
//           symbol histograms by LDS atomics
// and every 64 chunks (or at the end) closes a DEFLATE block:
//   codes   length-limited Huffman code lengths + canonical codes
//   emit    per-token bit strings, prefix sum of bit lengths, LDS-staged
//           OR-merge, coalesced dword stores
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "zgpu_dev.h"

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
constexpr int PEND = 3;             // searched items a wave may hold before it must parse the oldest
constexpr uint32_t LAG = 48;        // an item starts only when all items more than LAG back are parsed
constexpr uint32_t MAXDIST = 32768 - 64 * LAG - 128;  // window minus the span of in-flight items
constexpr uint32_t BLK_ITEMS = 1024;  // one DEFLATE block per 64 KiB of input
constexpr uint32_t LOOKAHEAD = 64 + 6 * 64 + 258 + 16;  // bytes an item may read past its start
constexpr uint32_t REFILL = 16384;  // ring refill granule
constexpr uint32_t CMP_CAP = 32;   // pass-1 compare length cap (bytes)
constexpr uint32_t SHARE_MIN = 16;  // matches at least this long offer their distance to the wave
constexpr int SHARE_ITERS = 6;      // distinct distances scored per wave
constexpr int STAGE_WORDS = 1792;   // bit staging (>= header + 1024*48/32 + carry)

// ---- LDS map (bytes) -------------------------------------------------------
constexpr uint32_t L_RING = 0;
constexpr uint32_t L_PREV = L_RING + RING + RING_PAD;      // u16[32768]
constexpr uint32_t L_HEAD = L_PREV + 65536;                // u16[1<<HBITS]
constexpr uint32_t L_FL = L_HEAD + (2u << HBITS);          // u32[288] lit/len freqs
constexpr uint32_t L_FD = L_FL + 288 * 4;                  // u32[32]  dist freqs
constexpr uint32_t L_MISC = L_FD + 32 * 4;                 // u32[64]  scan partials, control
constexpr uint32_t L_SCR = L_MISC + 64 * 4;                // phase scratch
constexpr uint32_t SCR_BYTES = 14336;
constexpr uint32_t L_TOTAL = L_SCR + SCR_BYTES;
static_assert(L_TOTAL <= 163840, "LDS budget");

// scratch, item loop (parse bookkeeping of in-flight items)
constexpr uint32_t NSLOT = 64;                       // > LAG
constexpr uint32_t S_EXIT = 0;                       // u16[NSLOT*64]
constexpr uint32_t S_CNT = S_EXIT + NSLOT * 64 * 2;  // u8[NSLOT*64]
constexpr uint32_t S_DONE = S_CNT + NSLOT * 64;      // u32[NSLOT]
constexpr uint32_t S_ENT = S_DONE + NSLOT * 4;       // u32[NSLOT]
constexpr uint32_t S_TOKB = S_ENT + NSLOT * 4;       // u32[NSLOT]
static_assert(S_TOKB + NSLOT * 4 <= SCR_BYTES, "item-loop scratch");
// scratch, block close phase
constexpr uint32_t S_STAGE = 0;                        // u32[STAGE_WORDS]
constexpr uint32_t S_LLTAB = S_STAGE + STAGE_WORDS * 4;  // u32[288] code | len<<16
constexpr uint32_t S_DTAB = S_LLTAB + 288 * 4;           // u32[32]
constexpr uint32_t S_HUF = S_DTAB + 32 * 4;              // Huffman build scratch
constexpr uint32_t HUF_BYTES = SCR_BYTES - S_HUF;
// Huffman build scratch map (bytes from S_HUF)
constexpr uint32_t H_KEY = 0;        // u32[512] sort keys
constexpr uint32_t H_PARENT = 2048;  // u16[640]
constexpr uint32_t H_IW = 3328;      // u32[288] internal weights / depths / first codes
constexpr uint32_t H_LENS = 4480;    // u8[288]
constexpr uint32_t H_CLF = 4768;     // u32[20] code-length symbol freqs
constexpr uint32_t H_CLTAB = 4848;   // u32[20]
constexpr uint32_t H_RLE = 4928;     // u16[320]
constexpr uint32_t H_NRLE = 5568;    // u32
static_assert(HUF_BYTES >= 5572, "huffman scratch");

// misc words
enum { M_SCAN = 0 /*[16]*/, M_UNIT = 32, M_NTOK = 34, M_TMP = 35, M_OVF = 36,
       M_ITEM = 40,    // next item to hand out
       M_LINK = 41,    // item whose turn it is to link its hash chains
       M_PARSE = 42,   // item whose turn it is to take the parse hand-off
       M_ENTRY = 43,   // unit position of the next token (belongs to item M_PARSE)
       M_LOADED = 44,  // input bytes resident in the ring (exclusive end)
       M_LOCK = 45     // try-lock of the parse frontier
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

__device__ inline uint32_t ld4u(const uint8_t *ring, uint32_t idx) {
    const uint32_t *w = (const uint32_t *)(ring + (idx & ~3u));
    uint32_t lo = w[0], hi = w[1];
    return __builtin_amdgcn_alignbyte(hi, lo, idx & 3u);
}

// input bytes [lo, hi) of the unit -> ring (and the mirror of its first bytes);
// lo is a multiple of 16.  `nthreads` threads with ids `t` cooperate.
__device__ inline void load_ring(uint8_t *ring, const uint8_t *in, uint32_t lo, uint32_t hi, int t,
                                 int nthreads, bool aligned) {
    uint32_t done = lo;
    if (aligned) {
        const uint32_t hi16 = lo + ((hi - lo) & ~15u);
        for (uint32_t p = lo + (uint32_t)t * 16; p < hi16; p += (uint32_t)nthreads * 16) {
            const uint4 v = *(const uint4 *)(in + p);
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

// unaligned dword from LDS in one instruction (gfx950 supports unaligned DS access)
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
__device__ inline uint32_t ld4(const uint8_t *p) { return *(const u32_unaligned *)p; }

// 16 bytes at any LDS address in one instruction, and the number of leading
// bytes (0..16) in which two such pieces agree
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef v4u __attribute__((aligned(1))) v4u_unaligned;
__device__ inline v4u ld16(const uint8_t *p) { return *(const v4u_unaligned *)p; }
__device__ inline uint32_t mlen16(v4u a, v4u b) {
    const v4u x = a ^ b;
    const uint64_t lo = (uint64_t)x.x | ((uint64_t)x.y << 32), hi = (uint64_t)x.z | ((uint64_t)x.w << 32);
    if (lo) return (uint32_t)__builtin_ctzll(lo) >> 3;
    if (hi) return 8u + ((uint32_t)__builtin_ctzll(hi) >> 3);
    return 16u;
}

__device__ inline uint32_t hash3(uint32_t w) {
    return ((w & 0xffffffu) * 0x9E3779B1u) >> (32 - HBITS);
}

// exclusive prefix sum over the workgroup; returns this thread's offset and
// the total through *total.  Uses misc[M_SCAN..+16].  Two barriers.
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
    for (int w = 0; w < WG / 64; ++w) {
        uint32_t p = part[w];
        if (w < wave) wsum += p;
        tot += p;
    }
    __syncthreads();
    *total = tot;
    return wsum + x - v;
}

// bitonic sort of n (power of two, <= 1024) u32 keys in LDS, ascending.
__device__ inline void bitonic_sort(uint32_t *a, int n) {
    const int t = threadIdx.x;
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (t < n) {
                int ixj = t ^ j;
                if (ixj > t) {
                    uint32_t x = a[t], y = a[ixj];
                    bool up = (t & k) == 0;
                    if ((x > y) == up) {
                        a[t] = y;
                        a[ixj] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// Wave-to-wave hand-off through one LDS word.  LDS operations of a wave are
// issued and completed in order and the LDS is a single coherent memory for
// the workgroup, so a wave that has drained its own LDS queue (lgkmcnt(0)) may
// publish, and a wave that has seen the value may read what was written before
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) lds_st(word, value);
}

// bitonic sort of one key per lane across the wave (ascending by lane), in registers
__device__ inline uint32_t wave_sort64(uint32_t key) {
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (uint32_t k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t other = __shfl_xor(key, (int)j, 64);
            const bool up = (lane & k) == 0;
            const bool lower = (lane & j) == 0;
            const uint32_t mn = key < other ? key : other, mx = key < other ? other : key;
            key = (lower == up) ? mn : mx;
        }
    }
    return key;
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
// freq[0..nsym) -> len[0..nsym), bit-reversed codes packed as code | len<<16 in
// tab[].  The workgroup sorts; thread 0 runs the two-queue tree build and the
// overflow repair.  At least two symbols get a code so every code set is
// complete.  scratch: >= 4096 bytes.
__device__ void build_codes(uint32_t *freq, int nsym, int maxbits, uint32_t *tab, uint8_t *scratch) {
    const int t = threadIdx.x;
    uint32_t *key = (uint32_t *)(scratch + H_KEY);
    uint16_t *parent = (uint16_t *)(scratch + H_PARENT);
    uint32_t *iw = (uint32_t *)(scratch + H_IW);
    uint8_t *lens = scratch + H_LENS;  // code length per symbol
    if (t == 0) {
        int nz = 0;
        for (int i = 0; i < nsym && nz < 2; ++i) nz += freq[i] != 0;
        // force two coded symbols (keeps the code complete)
        for (int i = 0; nz < 2 && i < nsym; ++i)
            if (freq[i] == 0) {
                freq[i] = 1;
                ++nz;
            }
    }
    __syncthreads();
    if (t < 512) key[t] = (t < nsym && freq[t]) ? ((freq[t] << 9) | (uint32_t)t) : 0xffffffffu;
    __syncthreads();
    bitonic_sort(key, 512);
    if (t < 288) lens[t] = 0;
    __syncthreads();
    if (t == 0) {
        int m = 0;
        while (m < nsym && key[m] != 0xffffffffu) ++m;
        // two-queue merge: leaves 0..m-1 (ascending weight), internal m..2m-2
        int li = 0, ii = 0;
        for (int k = 0; k < m - 1; ++k) {
            uint32_t w = 0;
            for (int r = 0; r < 2; ++r) {
                bool take_leaf;
                if (li >= m)
                    take_leaf = false;
                else if (ii >= k)
                    take_leaf = true;
                else
                    take_leaf = (key[li] >> 9) <= iw[ii];
                if (take_leaf) {
                    w += key[li] >> 9;
                    parent[li++] = (uint16_t)(m + k);
                } else {
                    w += iw[ii];
                    parent[m + ii++] = (uint16_t)(m + k);
                }
            }
            iw[k] = w;
        }
        // depths top-down (reuse iw as depth of internal nodes)
        uint32_t blc[32];
        for (int b = 0; b < 32; ++b) blc[b] = 0;
        iw[m - 2] = 0;
        for (int k = m - 3; k >= 0; --k) iw[k] = iw[parent[m + k] - m] + 1;
        int overflow = 0;
        for (int i = 0; i < m; ++i) {
            int d = (int)iw[parent[i] - m] + 1;
            if (d > maxbits) {
                d = maxbits;
                ++overflow;
            }
            blc[d]++;
        }
        // Kraft repair: move the deepest leaves up until the code fits
        while (overflow > 0) {
            int bits = maxbits - 1;
            while (blc[bits] == 0) --bits;
            blc[bits]--;
            blc[bits + 1] += 2;
            blc[maxbits]--;
            overflow -= 2;
        }
        // assign lengths in sorted order: rarest symbols get the longest codes
        int r = 0;
        for (int b = maxbits; b >= 1; --b)
            for (uint32_t c = 0; c < blc[b]; ++c) lens[key[r++] & 511] = (uint8_t)b;
        // canonical first codes per length
        uint32_t code = 0;
        uint32_t *next = iw;  // reuse [0..16)
        for (int b = 1; b <= maxbits; ++b) {
            code = (code + blc[b - 1]) << 1;
            next[b] = code;
        }
    }
    __syncthreads();
    // canonical code of symbol t = first[len] + #{s < t : len[s] == len[t]}
    if (t < nsym) {
        uint32_t l = lens[t], v = 0;
        if (l) {
            uint32_t rank = 0;
            for (int s = 0; s < t; ++s) rank += lens[s] == l;
            uint32_t c = iw[l] + rank;
            v = (__brev(c) >> (32 - l)) | (l << 16);
        }
        tab[t] = v;
    }
    __syncthreads();
}

struct Params {
    const uint8_t *in;
    const uint64_t *in_off;    // n_units+1
    const uint64_t *slot_off;  // n_units+1, 16-byte aligned slots in out_tmp
    uint8_t *out_tmp;
    uint32_t *out_size;  // n_units
    uint32_t *status;    // n_units
    uint32_t *tok;       // gridDim.x * TOK_CAP
    uint32_t *counter;   // unit queue head
    uint32_t n_units;
    int ctype, lazy, max_chain, nice;
    uint32_t flags;
    unsigned long long *prof;  // 16 phase counters (diagnostic builds) or null
};

}  // namespace

extern "C" __global__ void __launch_bounds__(WG)
zgpu_deflate_kernel(Params P) {
    __shared__ __align__(16) uint8_t lds_raw[L_TOTAL];
    Lds L{lds_raw};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t *ring = L.ring();
    uint16_t *prevt = L.prev();
    uint16_t *head = L.head();
    uint32_t *fl = L.fl(), *fd = L.fd(), *misc = L.misc();
    uint8_t *scr = L.scr();
    uint32_t *tokbuf = P.tok + (size_t)blockIdx.x * ZGPU_TOK_CAP;
    PH_DECL;

    for (;;) {
        if (tid == 0) misc[M_UNIT] = atomicAdd(P.counter, 1u);
        __syncthreads();
        const uint32_t u = misc[M_UNIT];
        __syncthreads();
        if (u >= P.n_units) break;

        const uint8_t *in = P.in + P.in_off[u];
        const uint32_t n = (uint32_t)(P.in_off[u + 1] - P.in_off[u]);
        uint32_t *out32 = (uint32_t *)(P.out_tmp + P.slot_off[u]);
        const uint32_t cap_words = (uint32_t)((P.slot_off[u + 1] - P.slot_off[u]) >> 2);
        const bool final_unit =
            (P.flags & 1u) || ((P.flags & 4u) && u == P.n_units - 1) || !(P.flags & 2u);
        const bool sync_unit = !final_unit;

        // ---- unit init ----
        for (int i = tid; i < (1 << HBITS) / 8; i += WG)
            ((uint4 *)head)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
        for (int i = tid; i < 320; i += WG) fl[i] = 0;  // fl[288] + fd[32] are adjacent
        uint32_t bitpos = 0;      // bits emitted so far in this unit (uniform)
        uint32_t base_word = 0;   // output word index of stage[0] (uniform)
        uint32_t carry_word = 0;  // partial word carried between blocks (uniform)
        const uint32_t nitems = (n + 63) >> 6;  // items of 64 positions
        const bool aligned = (((uintptr_t)in) & 15u) == 0;
        PH_START();
        {
            // first 32 KiB of the unit -> ring (all threads)
            const uint32_t hi = n < 2 * REFILL ? n : 2 * REFILL;
            load_ring(ring, in, 0, hi, tid, WG, aligned);
            if (tid == 0) {
                misc[M_ITEM] = 0;
                misc[M_LINK] = 0;
                misc[M_PARSE] = 0;
                misc[M_ENTRY] = 0;
                misc[M_NTOK] = 0;
                misc[M_OVF] = 0;
                misc[M_LOADED] = hi;
                misc[M_LOCK] = 0;
            }
        }
        __syncthreads();
        PH_END(0);

        // ---- close a DEFLATE block (all threads; uniform) ----
        auto close_block = [&](const bool last_block) {
            PH_START();
            // ---- close a DEFLATE block ----
            const uint32_t bfinal = (last_block && final_unit) ? 1u : 0u;
            uint32_t *stage = (uint32_t *)(scr + S_STAGE);
            uint32_t *lltab = (uint32_t *)(scr + S_LLTAB);
            uint32_t *dtab = (uint32_t *)(scr + S_DTAB);
            __syncthreads();
            if (tid == 0) {
                tokbuf[misc[M_NTOK]++] = 256;  // end of block
                fl[256]++;
            }
            for (int i = tid; i < STAGE_WORDS; i += WG) stage[i] = 0;
            __syncthreads();
            if (tid == 0) stage[0] = carry_word;
            BitW bw{stage, bitpos - base_word * 32};
            if (P.ctype == 2) {
                build_codes(fl, 286, 15, lltab, scr + S_HUF);
                build_codes(fd, 30, 15, dtab, scr + S_HUF);
                // ---- header: HLIT/HDIST, RLE of code lengths, code-length code ----
                uint32_t *clf = (uint32_t *)(scr + S_HUF + H_CLF);      // freqs of symbols 0..18
                uint32_t *cltab = (uint32_t *)(scr + S_HUF + H_CLTAB);
                uint16_t *rle = (uint16_t *)(scr + S_HUF + H_RLE);      // sym | extra<<8
                uint32_t *nrle = (uint32_t *)(scr + S_HUF + H_NRLE);
                if (tid < 20) clf[tid] = 0;
                __syncthreads();
                if (tid == 0) {
                    int hlit = 286, hdist = 30;
                    while (hlit > 257 && (lltab[hlit - 1] >> 16) == 0) --hlit;
                    while (hdist > 1 && (dtab[hdist - 1] >> 16) == 0) --hdist;
                    const int tot = hlit + hdist;
                    int nr = 0;
                    for (int i = 0; i < tot;) {
                        const uint32_t v = i < hlit ? lltab[i] >> 16 : dtab[i - hlit] >> 16;
                        int run = 1;
                        while (i + run < tot) {
                            const int k = i + run;
                            const uint32_t w = k < hlit ? lltab[k] >> 16 : dtab[k - hlit] >> 16;
                            if (w != v) break;
                            ++run;
                        }
                        i += run;
                        if (v == 0) {
                            while (run >= 11) {
                                int r = run > 138 ? 138 : run;
                                rle[nr++] = (uint16_t)(18 | ((r - 11) << 8));
                                clf[18]++;
                                run -= r;
                            }
                            if (run >= 3) {
                                rle[nr++] = (uint16_t)(17 | ((run - 3) << 8));
                                clf[17]++;
                                run = 0;
                            }
                            while (run-- > 0) {
                                rle[nr++] = 0;
                                clf[0]++;
                            }
                        } else {
                            rle[nr++] = (uint16_t)v;
                            clf[v]++;
                            --run;
                            while (run >= 3) {
                                int r = run > 6 ? 6 : run;
                                rle[nr++] = (uint16_t)(16 | ((r - 3) << 8));
                                clf[16]++;
                                run -= r;
                            }
                            while (run-- > 0) {
                                rle[nr++] = (uint16_t)v;
                                clf[v]++;
                            }
                        }
                    }
                    *nrle = (uint32_t)nr | ((uint32_t)hlit << 16) | ((uint32_t)hdist << 26);
                }
                __syncthreads();
                build_codes(clf, 19, 7, cltab, scr + S_HUF);
                if (tid == 0) {
                    const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5,
                                               11, 4, 12, 3, 13, 2, 14, 1, 15};
                    const uint32_t pk = *nrle;
                    const int nr = pk & 0xffff, hlit = (pk >> 16) & 0x3ff, hdist = pk >> 26;
                    int hclen = 19;
                    while (hclen > 4 && (cltab[order[hclen - 1]] >> 16) == 0) --hclen;
                    bw.put(bfinal, 1);
                    bw.put(2, 2);
                    bw.put((uint32_t)(hlit - 257), 5);
                    bw.put((uint32_t)(hdist - 1), 5);
                    bw.put((uint32_t)(hclen - 4), 4);
                    for (int i = 0; i < hclen; ++i) bw.put(cltab[order[i]] >> 16, 3);
                    for (int i = 0; i < nr; ++i) {
                        const uint32_t sym = rle[i] & 0xff, ex = rle[i] >> 8;
                        bw.put(cltab[sym] & 0xffff, cltab[sym] >> 16);
                        if (sym == 16) bw.put(ex, 2);
                        if (sym == 17) bw.put(ex, 3);
                        if (sym == 18) bw.put(ex, 7);
                    }
                    misc[M_TMP] = bw.rel;
                }
            } else {
                // fixed Huffman codes (RFC 1951 3.2.6; FixedHuffmanTable rawdeflate.js:48-68)
                if (tid < 288) {
                    uint32_t c, l;
                    if (tid <= 143) {
                        c = 0x30 + tid;
                        l = 8;
                    } else if (tid <= 255) {
                        c = 0x190 + tid - 144;
                        l = 9;
                    } else if (tid <= 279) {
                        c = tid - 256;
                        l = 7;
                    } else {
                        c = 0xc0 + tid - 280;
                        l = 8;
                    }
                    lltab[tid] = (__brev(c) >> (32 - l)) | (l << 16);
                }
                if (tid < 32) dtab[tid] = (__brev((uint32_t)tid) >> 27) | (5u << 16);
                if (tid == 0) {
                    bw.put(bfinal, 1);
                    bw.put(1, 2);
                    misc[M_TMP] = bw.rel;
                }
            }
            __syncthreads();
            bitpos = base_word * 32 + misc[M_TMP];
            const uint32_t ntok = misc[M_NTOK];
            __syncthreads();
            PH_END(5);
            // ---- emit tokens ----
            for (uint32_t t0 = 0; t0 < ntok; t0 += WG) {
                uint64_t bits = 0;
                uint32_t nb = 0;
                if (t0 + tid < ntok) {
                    const uint32_t tk = tokbuf[t0 + tid];
                    if (!(tk & 0x80000000u)) {
                        const uint32_t e = lltab[tk];
                        bits = e & 0xffff;
                        nb = e >> 16;
                    } else {
                        uint32_t sym, eb, ev;
                        len_sym((tk >> 16) & 0xff, sym, eb, ev);
                        uint32_t e = lltab[sym];
                        bits = e & 0xffff;
                        nb = e >> 16;
                        bits |= (uint64_t)ev << nb;
                        nb += eb;
                        dist_sym(tk & 0xffff, sym, eb, ev);
                        e = dtab[sym];
                        bits |= (uint64_t)(e & 0xffff) << nb;
                        nb += e >> 16;
                        bits |= (uint64_t)ev << nb;
                        nb += eb;
                    }
                }
                uint32_t total;
                const uint32_t off = block_scan_excl(nb, &total, misc + M_SCAN);
                if (nb) {
                    const uint32_t rel = bitpos + off - base_word * 32;
                    const uint32_t w = rel >> 5, sh = rel & 31;
                    const uint64_t lo = bits << sh;
                    atomicOr(&stage[w], (uint32_t)lo);
                    if ((uint32_t)(lo >> 32)) atomicOr(&stage[w + 1], (uint32_t)(lo >> 32));
                    if (sh && (nb + sh > 64)) atomicOr(&stage[w + 2], (uint32_t)(bits >> (64 - sh)));
                }
                __syncthreads();
                bitpos += total;
                const uint32_t full = (bitpos >> 5) - base_word;
                const bool fits = base_word + full <= cap_words;
                if (fits)
                    for (uint32_t i = tid; i < full; i += WG) out32[base_word + i] = stage[i];
                else if (tid == 0)
                    misc[M_OVF] = 1;
                __syncthreads();
                const uint32_t cw = stage[full];
                __syncthreads();
                for (uint32_t i = tid; i <= full + 2 && i < (uint32_t)STAGE_WORDS; i += WG) stage[i] = 0;
                __syncthreads();
                if (tid == 0) stage[0] = cw;
                base_word += full;
                __syncthreads();
            }
            PH_END(6);
            carry_word = stage[0];
            __syncthreads();
            // reset block state
            for (int i = tid; i < 320; i += WG) fl[i] = 0;
            if (tid == 0) misc[M_NTOK] = 0;
            __syncthreads();
                };

        if (n == 0) close_block(true);
        for (uint32_t blk_first = 0; blk_first < nitems; blk_first += BLK_ITEMS) {
            const uint32_t blk_end = blk_first + BLK_ITEMS < nitems ? blk_first + BLK_ITEMS : nitems;
            // ---- item loop: every wave takes the next 64 positions, no workgroup barrier ----
            // The parse of an item is deferred until the wave has searched its next item, so
            // that the in-order parse hand-off rarely has to wait for a slower earlier item.
            // ---- parse bookkeeping shared by the waves (in the phase scratch) ----
            // Per in-flight item (slot = item mod NSLOT): for every local index the exit
            // index and the number of tokens when the parse enters the item there, the
            // "searched" stamp, and -- once the parse frontier has passed the item -- its
            // entry position and token base.  Any wave moves the frontier (M_PARSE, M_ENTRY,
            // M_NTOK) under a try-lock; it never waits for the item's owner.
            uint16_t *exit_tab = (uint16_t *)(scr + S_EXIT);
            uint8_t *cnt_tab = scr + S_CNT;
            uint32_t *slot_done = (uint32_t *)(scr + S_DONE);
            uint32_t *slot_entry = (uint32_t *)(scr + S_ENT);
            uint32_t *slot_tokb = (uint32_t *)(scr + S_TOKB);
            if (tid < (int)NSLOT) slot_done[tid] = 0;  // the scratch held other data during block close
            __syncthreads();
            auto try_advance = [&]() {
                if (lane == 0) {
                    for (;;) {
                        if (atomicCAS(&misc[M_LOCK], 0u, 1u) != 0u) break;  // someone else is at it
                        uint32_t f = misc[M_PARSE], e = misc[M_ENTRY], t = misc[M_NTOK];
                        while (lds_ld(&slot_done[f & (NSLOT - 1)]) == f + 1) {
                            const uint32_t sl = f & (NSLOT - 1), le = e - (f << 6);
                            slot_entry[sl] = e;
                            slot_tokb[sl] = t;
                            if (le < 64u) {
                                e = (f << 6) + exit_tab[sl * 64 + le];
                                t += cnt_tab[sl * 64 + le];
                            }
                            ++f;
                        }
                        misc[M_ENTRY] = e;
                        misc[M_NTOK] = t;
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        lds_st(&misc[M_PARSE], f);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        lds_st(&misc[M_LOCK], 0u);
                        // an item stamped while the lock was held must not be left behind
                        if (lds_ld(&slot_done[f & (NSLOT - 1)]) != f + 1) break;
                    }
                }
            };
            // Up to PEND searched items wait per wave until the frontier has passed them
            // (slot 0 is the oldest); then their tokens are emitted.
            int npend = 0;
            uint32_t q_k[PEND], q_len[PEND], q_dist[PEND];
            // kind / successor of every local index of an item (0 literal, 1 match here,
            // 2 literal + match at +1); no look-ahead across the item boundary
            auto successor = [&](uint32_t best_len, uint32_t &kind) -> uint32_t {
                uint32_t l1 = __shfl_down(best_len, 1, 64);
                if (lane == 63) l1 = 0;
                uint32_t adv = 1;
                kind = 0;
                if (best_len >= 3) {
                    if (P.lazy > 0 && best_len < (uint32_t)P.lazy && l1 > best_len) {
                        kind = 2;
                        adv = 1 + l1;
                    } else {
                        kind = 1;
                        adv = best_len;
                    }
                }
                return (uint32_t)lane + adv;
            };
            auto resolved = [&](uint32_t k0) -> bool {
                return (int32_t)(lds_ld(&misc[M_PARSE]) - k0) > 0;
            };
            auto emit_oldest = [&]() {
                // ---- tokens of item q_k[0], in registers: lane l <-> local index l ----
                // five pointer-doubling steps rebuild the jump tables; lane j then follows j
                // hops from the entry by composing the tables along the bits of j, so the
                // tokens come out already compacted.
                const uint32_t k0 = q_k[0], best_len = q_len[0], best_dist = q_dist[0];
                const uint32_t p0 = k0 << 6, sl = k0 & (NSLOT - 1);
                const uint32_t seg_end = (n - p0) < 64u ? (n - p0) : 64u;
                PH_START();
                while (!resolved(k0)) {
                    try_advance();
                    __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
                PH_END(15);
                const uint32_t e = slot_entry[sl] - p0;  // local index of the first token
                const uint32_t tokbase = slot_tokb[sl];
                uint32_t kind;
                uint32_t QJ[6];
                QJ[0] = successor(best_len, kind);
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const uint32_t tj = __shfl(QJ[r], (int)(QJ[r] & 63u), 64);
                    QJ[r + 1] = QJ[r] < seg_end ? tj : QJ[r];
                }
                uint32_t Pk = e;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const uint32_t t2 = __shfl(QJ[r], (int)(Pk & 63u), 64);
                    if (((lane >> r) & 1) && Pk < seg_end) Pk = t2;
                }
                const bool on = Pk < seg_end;
                const uint32_t tk = (uint32_t)__shfl((int)kind, (int)(Pk & 63u), 64);
                const int mi = (int)((Pk + (tk == 2 ? 1u : 0u)) & 63u);
                const uint32_t tl = (uint32_t)__shfl((int)best_len, mi, 64);
                const uint32_t td = (uint32_t)__shfl((int)best_dist, mi, 64);
                const uint32_t cnt = on ? (tk == 2 ? 2u : 1u) : 0u;
                uint32_t inc = cnt;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t y = __shfl_up(inc, d, 64);
                    if (lane >= d) inc += y;
                }
                if (on) {
                    uint32_t *dst = tokbuf + tokbase + inc - cnt;
                    if (tk != 1) {
                        const uint32_t b = ring[(p0 + Pk) & (RING - 1)];
                        *dst++ = b;
                        atomicAdd(&fl[b], 1u);
                    }
                    if (tk != 0) {
                        *dst = 0x80000000u | (tl - 3) << 16 | (td - 1);
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
                    q_len[i] = q_len[i + 1];
                    q_dist[i] = q_dist[i + 1];
                }
                --npend;
                PH_END(7);
            };
            for (;;) {
                uint32_t k = 0;
                if (lane == 0) k = atomicAdd(&misc[M_ITEM], 1u);
                k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                if (k >= blk_end) break;
                PH_START();
                const uint32_t p0 = k << 6;
                const uint32_t p = p0 + (uint32_t)lane;
                // An unparsed item more than LAG items back must be parsed first, or the
                // waves that wait for it below could never be released.
                while (npend > 0 && k > q_k[0] + LAG) emit_oldest();
                // bound the spread of in-flight items: everything more than LAG items back is
                // parsed, so no wave still reads ring bytes or chain slots that get recycled
                if (k > LAG) lds_wait_ge(&misc[M_PARSE], k - LAG);
                // ring refill every 16 KiB, by the wave that crosses the boundary
                if (k && (p0 & (REFILL - 1)) == 0) {
                    uint32_t lo = p0 + REFILL, hi = p0 + 2 * REFILL;
                    if (hi > n) hi = n;
                    if (lo < hi) {
                        load_ring(ring, in, lo, hi, lane, 64, aligned);
                        lds_publish(&misc[M_LOADED], hi);
                    }
                }
                {
                    uint32_t need = p0 + LOOKAHEAD;
                    if (need > n) need = n;
                    lds_wait_ge(&misc[M_LOADED], need);
                }
#ifdef ZGPU_PROFILE
                if (tid == 0) ph_acc[3]++;
#endif
                PH_END(8);
                // ---- insert: sort (hash, lane) in registers, link chains in item order ----
                uint32_t key = 0xffffffffu;
                if (p + 3 <= n) key = (hash3(ld4(ring + (p & (RING - 1)))) << 6) | (uint32_t)lane;
                key = wave_sort64(key);
                {
                    const uint32_t kp = __shfl_up(key, 1, 64), kn = __shfl_down(key, 1, 64);
                    const uint32_t h = key >> 6;
                    const bool live = key != 0xffffffffu;
                    const bool first = lane == 0 || (kp >> 6) != h;
                    const bool last = lane == 63 || (kn >> 6) != h || kn == 0xffffffffu;
                    const uint32_t mypos = p0 + (key & 63u);
                    PH_END(9);
                    lds_wait_eq(&misc[M_LINK], k);
                    PH_END(10);
                    if (live) {
                        const uint32_t pp = first ? (uint32_t)head[h] : (p0 + (kp & 63u));
                        prevt[mypos & WMASK] = (uint16_t)pp;
                    }
                    if (live && last) head[h] = (uint16_t)mypos;
                    lds_publish(&misc[M_LINK], k + 1);
                }
                PH_END(11);
                // ---- search ----
                uint32_t best_len = 0, best_dist = 0;
                {
                    const uint32_t maxlen = p < n ? ((n - p) < 258u ? (n - p) : 258u) : 0u;
                    const uint32_t limit = p < MAXDIST ? p : MAXDIST;
                    uint32_t blen = 2;
                    {
                        // pass 1, lane per position: walk the hash chain.  A candidate costs one
                        // LDS round trip: its chain link and its first 16 bytes (one unaligned
                        // 128-bit read) are fetched together and compared against this position's
                        // 16 bytes held in registers, so there is no per-lane compare loop for the
                        // wave to diverge on.  Only a candidate that matches all 16 bytes reads 16
                        // more (CMP_CAP; pass 2 finds the exact length of long matches).
                        const uint32_t pi = p & (RING - 1);
                        const uint32_t cmpmax = maxlen < CMP_CAP ? maxlen : CMP_CAP;
                        const uint32_t stop = cmpmax < (uint32_t)P.nice ? cmpmax : (uint32_t)P.nice;
                        if (p + 3 <= n) {
                            uint32_t c = prevt[p & WMASK];
                            const v4u own = ld16(ring + pi);
                            uint32_t last = 0;
                            for (int budget = P.max_chain; budget > 0; --budget) {
                                const uint32_t dist = (p - c) & 0xffffu;
                                if (dist <= last || dist > limit) break;
                                last = dist;
                                const uint32_t ci = c & (RING - 1);
                                const uint32_t nxt = prevt[c & WMASK];
                                uint32_t len = mlen16(own, ld16(ring + ci));
                                if (len == 16 && cmpmax > 16)
                                    len = 16 + mlen16(ld16(ring + pi + 16), ld16(ring + ci + 16));
                                if (len > cmpmax) len = cmpmax;
                                if (len > blen) {
                                    blen = len;
                                    best_dist = dist;
                                    if (len >= stop) break;
                                }
                                c = nxt;
                            }
                        }
                    }
                    PH_END(12);
                    {
                        // pass 2, distance sharing across the wave: a distance that gave one
                        // lane a long match is scored for all 64 positions at once -- one byte
                        // compare per lane, a ballot, and count-trailing-ones give every lane
                        // its exact match length (up to 258) at that distance.
                        const uint32_t offer = best_dist;
                        unsigned long long todo = __ballot(blen >= SHARE_MIN);
                        for (int it = 0; it < SHARE_ITERS && todo; ++it) {
                            const int src = __builtin_ctzll(todo);
                            const uint32_t d = (uint32_t)__shfl((int)offer, src, 64);
                            uint32_t run = 0;
                            bool open = true;
                            for (uint32_t w = 0; w < 6; ++w) {
                                const uint32_t q = p0 + (w << 6) + (uint32_t)lane;
                                const bool eq = q < n && q >= d &&
                                                ring[q & (RING - 1)] == ring[(q - d) & (RING - 1)];
                                const unsigned long long m = __ballot(eq);
                                if (w == 0) {
                                    // zeros shifted in from the top read as mismatches after the inversion
                                    const unsigned long long inv = ~(m >> lane);
                                    run = inv ? (uint32_t)__builtin_ctzll(inv) : 64u;
                                    open = run == 64u - (uint32_t)lane;
                                } else if (open) {
                                    const uint32_t r2 = ~m ? (uint32_t)__builtin_ctzll(~m) : 64u;
                                    run += r2;
                                    open = r2 == 64u;
                                }
                                if (!__ballot(open && run < 258u)) break;
                            }
                            if (run > maxlen) run = maxlen;
                            if (run >= 3 && d <= limit && (run > blen || (run == blen && d < best_dist))) {
                                blen = run;
                                best_dist = d;
                            }
                            todo &= ~__ballot(offer == d);
                        }
                    }
                    if (blen >= 3 && !(blen == 3 && best_dist > 4096)) best_len = blen;
                }
                PH_END(13);
                PH_START();
                {
                    // ---- exit index and token count of every local index (six doubling steps) ----
                    const uint32_t seg_end = (n - p0) < 64u ? (n - p0) : 64u;
                    uint32_t kind;
                    uint32_t J = successor(best_len, kind);
                    if ((uint32_t)lane >= seg_end) J = (uint32_t)lane;  // past the end: stay put
                    uint32_t Cn = (uint32_t)lane < seg_end ? (kind == 2 ? 2u : 1u) : 0u;
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        const int idx = (int)(J & 63u);
                        const uint32_t tj = __shfl(J, idx, 64), tc = __shfl(Cn, idx, 64);
                        const bool inb = J < seg_end;
                        Cn += inb ? tc : 0u;
                        J = inb ? tj : J;
                    }
                    const uint32_t sl = k & (NSLOT - 1);
                    exit_tab[sl * 64 + lane] = (uint16_t)J;
                    cnt_tab[sl * 64 + lane] = (uint8_t)Cn;
                }
                PH_END(14);
                // tokens of items the frontier has passed leave before this item is stamped:
                // until then the frontier cannot run far enough to recycle their slots
                while (npend > 0 && (npend == PEND || resolved(q_k[0]))) emit_oldest();
#pragma unroll
                for (int i = 0; i < PEND; ++i)
                    if (i == npend) {
                        q_k[i] = k;
                        q_len[i] = best_len;
                        q_dist[i] = best_dist;
                    }
                ++npend;
                lds_publish(&slot_done[k & (NSLOT - 1)], k + 1);
                try_advance();
            }
            while (npend > 0) emit_oldest();
            PH_START();
            __syncthreads();
            PH_END(1);
            if (tid == 0) misc[M_ITEM] = blk_end;  // the counter overshoots by one per wave
            close_block(blk_end == nitems);
        }

        // ---- finish the unit ----
        PH_FLUSH();
#ifdef ZGPU_PROFILE
        for (int i_ = 0; i_ < 16; ++i_) ph_acc[i_] = 0;
#endif
        if (tid == 0) {
            uint32_t bp = bitpos, cw = carry_word, bwd = base_word;
            bool ovf = misc[M_OVF] != 0;
            auto put = [&](uint32_t v, uint32_t nb) {
                // serial writer straight to the output (at most a few words)
                while (nb) {
                    uint32_t sh = bp & 31, take = 32 - sh < nb ? 32 - sh : nb;
                    cw |= (v & (take == 32 ? ~0u : ((1u << take) - 1))) << sh;
                    v = take == 32 ? 0 : v >> take;
                    nb -= take;
                    bp += take;
                    if ((bp & 31) == 0) {
                        if (bwd < cap_words)
                            out32[bwd] = cw;
                        else
                            ovf = true;
                        ++bwd;
                        cw = 0;
                    }
                }
            };
            if (sync_unit) {
                put(0, 3);  // BFINAL=0, BTYPE=00
                put(0, (8 - (bp & 7)) & 7);
                put(0x0000, 16);
                put(0xffff, 16);
            } else
                put(0, (8 - (bp & 7)) & 7);
            if (bp & 31) {
                if (bwd < cap_words)
                    out32[bwd] = cw;
                else
                    ovf = true;
            }
            P.out_size[u] = bp >> 3;
            P.status[u] = ovf ? 1u : 0u;
        }
        __syncthreads();
    }
}

// ---- stored blocks (CompressionType.NONE; makeNocompressBlock rawdeflate.js:131-153)
extern "C" __global__ void __launch_bounds__(256)
zgpu_stored_kernel(const uint8_t *in, const uint64_t *in_off, const uint64_t *slot_off,
                   uint8_t *out_tmp, uint32_t *out_size, uint32_t *status, uint32_t n_units,
                   uint32_t flags) {
    for (uint32_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint8_t *src = in + in_off[u];
        const uint64_t n = in_off[u + 1] - in_off[u];
        uint8_t *dst = out_tmp + slot_off[u];
        const bool final_unit = (flags & 1u) || ((flags & 4u) && u == n_units - 1) || !(flags & 2u);
        const uint64_t nblk = n ? (n + 65534) / 65535 : 1;
        for (uint64_t b = 0; b < nblk; ++b) {
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
            for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) d[5 + i] = src[o + i];
        }
        if (threadIdx.x == 0) {
            out_size[u] = (uint32_t)(n + 5 * nblk);
            status[u] = 0;
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
