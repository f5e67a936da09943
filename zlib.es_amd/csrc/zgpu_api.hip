// Host side of libzgpu.so: the C ABI declared in include/zgpu.h.
// Thin by design -- unit tables, workspace, launches and copies.  No CPU
// compression or checksum path exists here: without a GPU every entry point
// returns ZGPU_E_NODEVICE.
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <queue>
#include <vector>

#include "../../include/zgpu.h"
#include "zgpu_dev.h"
#include "deflate_kernel.hip"
#include "checksum_kernel.hip"

namespace {

thread_local int g_last_hip = 0;
// Engine knobs: process-wide, read once when a plan is set up (a plan keeps what it read).
std::atomic<int> g_max_chain{8};
std::atomic<int> g_nice{32};
std::atomic<uint64_t> g_split_bytes{1ull << 20};  // units longer than twice this are cut into pieces of at most this size
std::atomic<int> g_deep{96};       // long-match steps of the search kernel: deep-walk budget (0 = off)
// units of the "dense" content class (zgpu_classify_kernel) are searched with this chain budget (a
// thirty-second of it for those of many byte values), nice_length 258 and a lazy value that follows the
// class; 0 = no such pass
std::atomic<int> g_dense_chain{2048};
std::atomic<int> g_min_lazy{258};  // the parse always looks one position ahead unless told otherwise
std::atomic<int> g_live{0};        // plans and jobs alive: zgpu_shutdown leaves the device state alone while any is
unsigned long long *g_prof = nullptr;  // diagnostic builds only

#define HIPCHK(expr)                         \
    do {                                     \
        hipError_t e_ = (expr);              \
        if (e_ != hipSuccess) {              \
            g_last_hip = (int)e_;            \
            return e_ == hipErrorOutOfMemory ? ZGPU_E_NOMEM : ZGPU_E_NODEVICE; \
        }                                    \
    } while (0)

// ---- GF(2) arithmetic for CRC-32 (reflected, bit 31 = x^0) -----------------
constexpr uint32_t kPoly = 0xEDB88320u;
uint32_t h_gf_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? kPoly : 0u);
    }
    return p;
}
// x^(8*nbytes) mod P
uint32_t h_xpow_bytes(uint64_t nbytes) {
    uint32_t r = 0x80000000u;       // x^0
    uint32_t sq = 0x00800000u;      // x^8
    while (nbytes) {
        if (nbytes & 1) r = h_gf_mul(r, sq);
        sq = h_gf_mul(sq, sq);
        nbytes >>= 1;
    }
    return r;
}

struct HostCtx;
void free_host_ctx(HostCtx *c);
struct CkScratch;  // segment tables of zgpu_checksum_device (defined with DevBuf below)
void free_ck_scratch(CkScratch *c);

struct DeviceState {
    int device = -1;
    int cus = 0;
    uint32_t *d_tables = nullptr;  // checksum kernel tables
    uint32_t *d_tables_wide = nullptr;  // 11-bit tables of the wide checksum kernel
    CkScratch *ck = nullptr;       // scratch of zgpu_checksum_device
    HostCtx *host = nullptr;       // staging of the host-buffer entry points
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ck_ms = 0.f;
    std::mutex ck_mu;  // zgpu_checksum_device calls on one device share the scratch above
};
std::mutex g_mu;
std::vector<DeviceState *> g_devs;
std::vector<int> g_allowed;  // devices named by zgpu_init

constexpr size_t kCkTabWords = 20 * 256 + 8;  // what zgpu_checksum_kernel reads; the wave variant's block follows
void build_tables(std::vector<uint32_t> &t) {
    t.assign(kCkTabWords + 4 * 256, 0);
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (kPoly ^ (c >> 1)) : (c >> 1);
        t[i] = c;
    }
    for (int k = 1; k < 16; ++k)
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t p = t[(k - 1) * 256 + i];
            t[k * 256 + i] = (p >> 8) ^ t[p & 0xff];
        }
    const uint32_t adv = h_xpow_bytes(16ull * ZGPU_CK_THREADS);
    for (int j = 0; j < 4; ++j)
        for (uint32_t v = 0; v < 256; ++v) t[(16 + j) * 256 + v] = h_gf_mul(v << (8 * j), adv);
    for (int lv = 0; lv < 8; ++lv) t[20 * 256 + lv] = h_xpow_bytes(16ull << lv);
    // zgpu_checksum_wave_kernel: 64 lanes share a segment, a lane's remainder advances by 16 * 64 bytes
    const uint32_t adv64 = h_xpow_bytes(16ull * 64);
    for (int j = 0; j < 4; ++j)
        for (uint32_t v = 0; v < 256; ++v) t[kCkTabWords + j * 256 + v] = h_gf_mul(v << (8 * j), adv64);
}

// tables of zgpu_checksum_wide_kernel, derived from the byte-sliced ones by linearity
void build_tables_wide(const std::vector<uint32_t> &t, std::vector<uint32_t> &w) {
    w.assign(15 * 2048 + ZGPU_CKW_XPOW + 256, 0);
    // remainder contributed by bit j of a 16-byte piece: byte j/8 followed by 15 - j/8 more bytes
    uint32_t unit[128];
    for (int j = 0; j < 128; ++j) unit[j] = t[(15 - j / 8) * 256 + (1u << (j % 8))];
    for (int k = 0; k < 12; ++k)
        for (uint32_t v = 0; v < 2048; ++v) {
            uint32_t x = 0;
            for (int b = 0; b < 11 && 11 * k + b < 128; ++b)
                if (v >> b & 1) x ^= unit[11 * k + b];
            w[k * 2048 + v] = x;
        }
    // from the end of a lane's last piece of one round to the end of its first piece of the next
    const uint32_t adv = h_xpow_bytes(16ull * ((uint64_t)ZGPU_CKW_PIECES * ZGPU_CKW_THREADS - ZGPU_CKW_PIECES + 1));
    for (int c = 0; c < 3; ++c)
        for (uint32_t v = 0; v < 2048; ++v) {
            const uint64_t sh = (uint64_t)v << (11 * c);
            w[(12 + c) * 2048 + v] = sh >> 32 ? 0u : h_gf_mul((uint32_t)sh, adv);
        }
    for (int lv = 0; lv < ZGPU_CKW_XPOW; ++lv) w[15 * 2048 + lv] = h_xpow_bytes(16ull << lv);
    for (int i = 0; i < 256; ++i) w[15 * 2048 + ZGPU_CKW_XPOW + i] = t[i];
}

void free_state(DeviceState *s) {
    (void)hipFree(s->d_tables);
    (void)hipFree(s->d_tables_wide);
    free_ck_scratch(s->ck);
    free_host_ctx(s->host);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
}

// The host-buffer entry points run on the current device, or, when zgpu_init named
// devices and the current one is not among them, on the first named; the caller's
// current device is restored on return.
struct HostDeviceScope {
    int prev = -1;
    int rc = ZGPU_OK;
    HostDeviceScope() {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) {
            rc = ZGPU_E_NODEVICE;
            return;
        }
        int want = cur;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            if (!g_allowed.empty()) {
                want = g_allowed[0];
                for (int d : g_allowed)
                    if (d == cur) want = cur;
            }
        }
        if (want != cur) {
            if (hipSetDevice(want) != hipSuccess) {
                rc = ZGPU_E_NODEVICE;
                return;
            }
            prev = cur;
        }
    }
    ~HostDeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int get_state(DeviceState **out) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto *s : g_devs)
        if (s->device == dev) {
            *out = s;
            return ZGPU_OK;
        }
    auto *s = new (std::nothrow) DeviceState();
    if (!s) return ZGPU_E_NOMEM;
    s->device = dev;
    const int rc = [&]() -> int {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, dev));
        s->cus = prop.multiProcessorCount;
        std::vector<uint32_t> t;
        build_tables(t);
        HIPCHK(hipMalloc(&s->d_tables, t.size() * 4));
        HIPCHK(hipMemcpy(s->d_tables, t.data(), t.size() * 4, hipMemcpyHostToDevice));
        std::vector<uint32_t> tw;
        build_tables_wide(t, tw);
        HIPCHK(hipMalloc(&s->d_tables_wide, tw.size() * 4));
        HIPCHK(hipMemcpy(s->d_tables_wide, tw.data(), tw.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipEventCreate(&s->ev0));
        HIPCHK(hipEventCreate(&s->ev1));
        return ZGPU_OK;
    }();
    if (rc) {
        free_state(s);
        return rc;
    }
    g_devs.push_back(s);
    *out = s;
    return ZGPU_OK;
}

uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// A device buffer that only grows: the workspace of a plan and the staging of the host-buffer
// entry points are kept from call to call, so a steady caller stops allocating.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return ZGPU_OK;
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;  // headroom: batches of a steady caller vary a little
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            want = bytes;
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) {
            g_last_hip = (int)e;
            p = nullptr;
            return e == hipErrorOutOfMemory ? ZGPU_E_NOMEM : ZGPU_E_NODEVICE;
        }
        cap = want;
        return ZGPU_OK;
    }
    void release() {
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const { return (T *)p; }
};

// the same for page-locked host memory (the device reads and writes it at full PCIe rate)
struct HostBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return ZGPU_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e != hipSuccess) {
            g_last_hip = (int)e;
            p = nullptr;
            return ZGPU_E_NOMEM;
        }
        cap = bytes;
        return ZGPU_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct CkScratch {
    DevBuf seg_off, seg_crc, seg_adler, out2;
};
void free_ck_scratch(CkScratch *c) {
    if (!c) return;
    c->seg_off.release();
    c->seg_crc.release();
    c->seg_adler.release();
    c->out2.release();
    delete c;
}

// the engine knobs as one plan sees them (read once, when the plan is set up)
struct Knobs {
    int max_chain, nice, deep, min_lazy;
    uint64_t split;
    int dense_chain;
};
Knobs read_knobs() {
    return Knobs{g_max_chain.load(), g_nice.load(), g_deep.load(), g_min_lazy.load(), g_split_bytes.load(), g_dense_chain.load()};
}

}  // namespace

struct zgpu_plan {
    DeviceState *ds = nullptr;
    uint32_t n_units = 0;
    int ctype = 2, lazy = 0;
    uint32_t flags = 0;
    Knobs knobs{};
    uint64_t in_total = 0, out_bound = 0;
    uint32_t n_pieces = 0;  // what the kernels see: units, the long ones cut into pieces
    bool split = false;     // some unit has more than one piece
    bool direct = false;    // the emit kernel writes units at their final offsets (no staging, no gathering pass)
    bool search_by_class = false;  // the search kernel's small-window variants take the units that fit them
    // Unit tables: built on the host into one blob, uploaded with one copy.  The pointers below
    // point into d_tab.
    std::vector<uint8_t> h_tab;
    DevBuf d_tab;
    uint64_t *d_in_off = nullptr, *d_slot_off = nullptr, *d_tok_off = nullptr, *d_unit_in_off = nullptr;
    uint32_t *d_blk_first = nullptr, *d_prime = nullptr, *d_first_piece = nullptr;
    uint8_t *d_final = nullptr;
    // per-unit checksums of long units: the units cut into segments of at most ZGPU_CK_SEG bytes
    // (so that many workgroups share a long unit), folded per unit afterwards
    uint32_t n_segs = 0;               // 0: every unit is its own segment
    uint64_t *d_seg_off = nullptr;     // n_segs + 1
    uint32_t *d_first_seg = nullptr;   // n_units + 1
    DevBuf b_seg_crc, b_seg_adler;
    // results and workspace
    DevBuf b_out_off, b_unit_out_off, b_out_size, b_status, b_crc, b_adler, b_counter, b_tok, b_hist, b_ntok, b_state, b_out_tmp;
    uint64_t *d_out_off() const { return b_out_off.as<uint64_t>(); }
    uint64_t *d_unit_out_off() const { return b_unit_out_off.as<uint64_t>(); }
    uint32_t *d_out_size() const { return b_out_size.as<uint32_t>(); }
    uint32_t *d_status() const { return b_status.as<uint32_t>(); }
    uint32_t *d_crc() const { return b_crc.as<uint32_t>(); }
    uint32_t *d_adler() const { return b_adler.as<uint32_t>(); }
    uint32_t *d_counter() const { return b_counter.as<uint32_t>(); }
    // cls_len[c] > 0: the units of size class c (0: up to kQuarterMax, 1: up to kHalfWindowMax, 2: longer) are
    // listed at d_list + cls_at[c]; all zero: the round is of one class (by max_bytes) and runs as a range
    struct Round { uint32_t first, end; uint64_t max_bytes; uint32_t cls_at[3] = {0, 0, 0}, cls_len[3] = {0, 0, 0}; };
    uint32_t *d_list = nullptr;
    // units of the dense content class, found on the device when the plan runs: b_dense is a list of the
    // largest round's worth of unit numbers (the tail 0xffffffff = no unit); its queue head and count are
    // the two words of b_counter from 6 * rounds on
    DevBuf b_dense;
    uint32_t dense_cap = 0;  // 0: the plan has no such pass
    // ... and one byte per unit (0, or 1 + its class) that says so to the first search, which passes those
    // units over: the two searches run side by side, the first on a stream of the plan's own that forks from
    // the caller's after the classification and joins it again before the emit kernel
    DevBuf b_taken;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::vector<Round> rounds;  // units whose tokens share the token buffer at one time
    std::vector<uint32_t> h_status;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ran = false;
    bool pooled = false;  // owned by a device's host context (not counted as a caller's live plan)
    void release_buffers() {
        for (DevBuf *b : {&d_tab, &b_out_off, &b_unit_out_off, &b_out_size, &b_status, &b_crc, &b_adler, &b_counter, &b_tok,
                          &b_hist, &b_ntok, &b_state, &b_out_tmp, &b_seg_crc, &b_seg_adler, &b_dense, &b_taken})
            b->release();
    }
};

// per-unit CRC-32 / Adler-32 from the values of a unit's segments (one thread per unit; same sums as above)
extern "C" __global__ void __launch_bounds__(256)
zgpu_unit_fold_kernel(const uint64_t *seg_off, const uint32_t *first_seg, const uint32_t *seg_crc,
                      const uint32_t *seg_adler, uint32_t n_units, uint32_t *crc_out, uint32_t *adler_out) {
    const uint32_t u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n_units) return;
    const uint32_t M = 65521u;
    const uint32_t s0 = first_seg[u], s1 = first_seg[u + 1];
    const uint64_t end = seg_off[s1];
    uint32_t crc = 0;
    unsigned long long a1 = 0, a2 = 0;
    for (uint32_t i = s0; i < s1; ++i) {
        const uint64_t after = end - seg_off[i + 1];
        uint32_t r = 0x80000000u, sq = 0x00800000u;
        for (uint64_t e = after; e; e >>= 1) {
            if (e & 1) r = gf_mul(r, sq);
            sq = gf_mul(sq, sq);
        }
        crc ^= gf_mul(seg_crc[i], r);
        const uint32_t a = seg_adler[i];
        const unsigned long long d1 = ((a & 0xffffu) + M - 1) % M;
        a1 += d1;
        a2 += (a >> 16) + d1 * (after % M) % M;
    }
    crc_out[u] = crc;
    adler_out[u] = (uint32_t)((a2 % M) << 16 | (1 + a1) % M);
}

namespace {
uint64_t unit_blocks(uint64_t n) { return n ? (n + ZGPU_BLOCK_BYTES - 1) / ZGPU_BLOCK_BYTES : 1; }
// token slots (16 bits each) reserved for a unit: full blocks at their fixed stride, the last
// one exact, rounded so that every region starts on 8 bytes and reads of four slots stay inside
uint64_t unit_tok_cap(uint64_t n) {
    const uint64_t nb = unit_blocks(n), last = n - (nb - 1) * ZGPU_BLOCK_BYTES;
    return (nb - 1) * ZGPU_BLOCK_TOK + ((last + 1 + 3) & ~3ull) + 4;
}

// ---- which kernel variant serves which units (thresholds read once) ----------------------
// the search kernel's half-window variant (two workgroups per CU) serves rounds of units of at most 16 KiB;
// ZGPU_HALF_WINDOW=1 / 0 forces it on / off for measurements
const int kHalfWindow = [] {
    const char *e = getenv("ZGPU_HALF_WINDOW");
    return e ? (e[0] == '1' ? 1 : 0) : -1;
}();
// units of at most this many bytes take the half-window variant (window 14,464 bytes; ZGPU_HALF_MAX tunes)
const uint64_t kHalfWindowMax = [] {
    const char *e = getenv("ZGPU_HALF_MAX");
    return e ? strtoull(e, nullptr, 10) : 16384ull;
}();
// ... and units of at most this many bytes the four-workgroups-per-CU variant, whose window is 7040 bytes
// (ZGPU_QUARTER_MAX tunes)
// under FIXED, units of at most this many bytes -- those that the half- and quarter-window variants serve -- are
// keyed on three bytes in the first search (ZGPU_KEY3_MAX tunes; 0 = never): real files cut to 300-3000 bytes were
// 1.015-1.036 x the reference's size with a third to all of them above 1.03 x, and are 0.994-1.003 x with none;
// files of 8000 and 16000 bytes 1.00-1.02 -> 0.98-1.00 x; the zip workload 0.9 % smaller at the same speed
const uint32_t kKey3Max = [] {
    const char *e = getenv("ZGPU_KEY3_MAX");
    return e ? (uint32_t)strtoul(e, nullptr, 10) : 16384u;
}();
// (ZGPU_KEY3_DYNAMIC=1: the same under DYNAMIC, for measurements)
const bool kKey3Dynamic = [] {
    const char *e = getenv("ZGPU_KEY3_DYNAMIC");
    return e && e[0] == '1';
}();
const uint64_t kQuarterMax = [] {
    const char *e = getenv("ZGPU_QUARTER_MAX");
    return e ? strtoull(e, nullptr, 10) : 8192ull;
}();
// rounds whose units are all at most this long take the wave-per-unit emit kernel (ZGPU_EMIT_SMALL_MAX tunes)
const uint64_t kEmitSmallMax = [] {
    const char *e = getenv("ZGPU_EMIT_SMALL_MAX");
    return e ? strtoull(e, nullptr, 10) : (16ull << 10);
}();

// (Re)targets a plan at a unit table: builds the piece / slot / token tables, grows the
// workspace where it is too small and uploads the tables on `st`.  history: bytes of the same
// stream that lie in the input buffer right before in_off[0] (they prime the first unit's window
// when the batch is a chunked stream with ZGPU_F_PRIME_WINDOW).
int plan_setup(zgpu_plan *p, const uint64_t *in_off, uint32_t n_units, int ctype, int lazy, uint32_t flags,
               uint64_t history, hipStream_t st, bool async) {
    if (!in_off || n_units == 0) return ZGPU_E_ARG;
    if (ctype < 0 || ctype > 2 || lazy < 0) return ZGPU_E_ARG;
    for (uint32_t i = 0; i < n_units; ++i) {
        if (in_off[i + 1] < in_off[i]) return ZGPU_E_ARG;
        if (in_off[i + 1] - in_off[i] > ZGPU_MAX_UNIT_BYTES) return ZGPU_E_TOOBIG;
    }
    DeviceState *ds = p->ds;
    p->n_units = n_units;
    p->ctype = ctype;
    p->lazy = lazy;
    p->flags = flags;
    p->knobs = read_knobs();
    p->in_total = in_off[n_units];
    p->ran = false;
    p->rounds.clear();
    // ---- pieces: what the kernels work on.  A unit longer than twice the split size is cut
    // into pieces of that size, each compressed by its own workgroup as a chunk of the unit's
    // stream: byte-aligned end (empty stored block), window primed with the 32 KiB before it.
    // A unit's pieces are consecutive, so its output is their concatenation.
    const bool chunked = (flags & ZGPU_F_SYNC_ALIGN) && !(flags & ZGPU_F_FINAL_EACH);
    const bool prime_units = chunked && (flags & ZGPU_F_PRIME_WINDOW);
    // Piece size: a batch with fewer large units than kPiecesPerCu per CU (the unit queue then balances
    // poorly at its tail, and the emit kernel's workgroups run short) cuts units longer than two
    // pieces into pieces of (their bytes) / (kPiecesPerCu x CUs), between 128 KiB and the set limit.
    constexpr uint32_t kPiecesPerCu = 16;
    uint64_t piece = 0;
    // Counted over the units above 16 KiB: the smaller ones run in the small-window kernel variants, several
    // to a CU, and do not help to fill the CUs the large ones leave idle.
    uint64_t n_big = 0, big_bytes = 0;
    for (uint32_t i = 0; i < n_units; ++i) {
        const uint64_t n = in_off[i + 1] - in_off[i];
        if (n > kHalfWindowMax) {
            ++n_big;
            big_bytes += n;
        }
    }
    if (ctype != ZGPU_NONE && p->knobs.split && n_big && n_big < kPiecesPerCu * (uint64_t)ds->cus) {
        piece = align_up(big_bytes / ((uint64_t)kPiecesPerCu * (uint64_t)ds->cus) + 1, 65536);
        if (piece < (128u << 10)) piece = 128u << 10;
        if (piece > p->knobs.split) piece = p->knobs.split;
        // Cutting is not free (a piece links the 32 KiB before it again and ends with its own blocks: about
        // 8 % on 256 KiB pieces), so it has to win on the clock: both ways are played through -- the units,
        // or their pieces, handed in order to the workgroup that is free first, time taken as bytes: one
        // search workgroup per CU, then eight emit workgroups per CU at 0.3 of the search's time per byte --
        // and the units stay whole unless the pieces finish sooner.  (1024 chunks of 1 MiB on 256 CUs are four
        // rounds of whole units; 600 such chunks are three rounds whole and 2.5 cut; 256 units of 4 MiB are
        // cut because whole they would keep 256 of the 2048 emit workgroups busy for 64 blocks each.)
        auto makespan = [&](bool cut) {
            uint64_t total = 0;
            for (int phase = 0; phase < 2; ++phase) {  // 0: search, 1: emit
                std::priority_queue<uint64_t, std::vector<uint64_t>, std::greater<uint64_t>> free_at;
                const int workers = phase ? ds->cus * ZGPU_EMIT_WG_PER_CU : ds->cus;
                for (int c = 0; c < workers; ++c) free_at.push(0);
                uint64_t end = 0;
                auto give = [&](uint64_t cost) {
                    const uint64_t t = free_at.top() + cost;
                    free_at.pop();
                    free_at.push(t);
                    if (t > end) end = t;
                };
                for (uint32_t i = 0; i < n_units; ++i) {
                    const uint64_t n = in_off[i + 1] - in_off[i];
                    if (n <= kHalfWindowMax) continue;
                    if (cut && n > 2 * piece) {
                        for (uint64_t o = 0; o < n; o += piece) {
                            const uint64_t len = n - o < piece ? n - o : piece;
                            give(phase ? len * 3 / 10 : len + (o ? 21000 : 0));  // priming and framing of a piece
                        }
                    } else
                        give(phase ? n * 3 / 10 : n);
                }
                total += end;
            }
            return total;
        };
        if (makespan(false) <= makespan(true)) piece = 0;
    }
    std::vector<uint64_t> pc_off;       // piece offsets into the input, n_pieces + 1
    std::vector<uint32_t> first_piece;  // per unit, n_units + 1
    std::vector<uint32_t> pc_prime;     // bytes of the same stream before the piece that prime its window
    std::vector<uint8_t> pc_final;      // piece ends its stream (BFINAL) / ends with an empty stored block
    for (uint32_t i = 0; i < n_units; ++i) {
        const uint64_t o = in_off[i], n = in_off[i + 1] - o;
        const uint64_t split = (piece && n > 2 * piece) ? piece : 0;
        const uint64_t np = split ? (n + split - 1) / split : 1;
        const bool unit_final = !chunked || ((flags & ZGPU_F_FINAL_LAST) && i == n_units - 1);
        first_piece.push_back((uint32_t)pc_off.size());
        for (uint64_t k = 0; k < np; ++k) {
            pc_off.push_back(o + k * split);
            uint64_t before = k ? k * split : (prime_units ? o - in_off[0] + history : 0);
            if (before > 32768) before = 32768;
            pc_prime.push_back((uint32_t)before & ~63u);
            pc_final.push_back(unit_final && k == np - 1);
        }
        if (pc_off.size() > 0x7fffffffull) return ZGPU_E_TOOBIG;
    }
    const uint32_t n_pieces = (uint32_t)pc_off.size();
    first_piece.push_back(n_pieces);
    pc_off.push_back(in_off[n_units]);
    p->n_pieces = n_pieces;
    p->split = n_pieces != n_units;
    std::vector<uint64_t> slot(n_pieces + 1);
    uint64_t acc = 0;
    // Stored blocks have sizes that follow from the unit sizes, so their final offsets are ready here.
    // For the Huffman modes the slots are bounds: what a caller must provide to be sure of room, and
    // -- for plans that stage their output (see p->direct) -- where a unit goes before it is placed.
    for (uint32_t i = 0; i < n_pieces; ++i) {
        slot[i] = acc;
        const uint64_t n = pc_off[i + 1] - pc_off[i];
        acc += ctype == ZGPU_NONE ? n + 5 * (n ? (n + 65534) / 65535 : 1) : align_up(zgpu_deflate_bound(n) + 32, 16);
    }
    slot[n_pieces] = acc;
    p->out_bound = acc;
    // Token scratch: 2 bytes per input byte in the worst case.  Pieces are cut into rounds
    // whose tokens fit a buffer of at most a third of the free device memory (ZGPU_TOK_BYTES
    // overrides; never less than one maximum-size piece needs).
    std::vector<uint64_t> tok_off(n_pieces);
    std::vector<uint32_t> blk_first(n_pieces);
    uint64_t tok_buf_tokens = 0, n_blocks = 0;
    if (ctype != ZGPU_NONE) {
        size_t free_b = 0, total_b = 0;
        uint64_t cap_bytes = 8ull << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap_bytes = (free_b + p->b_tok.cap) / 3;
        if (const char *e = getenv("ZGPU_TOK_BYTES")) cap_bytes = strtoull(e, nullptr, 10);
        const uint64_t cap_tokens = cap_bytes / 2;
        uint64_t used = 0, round_max = 0;
        uint32_t first = 0;
        for (uint32_t i = 0; i < n_pieces; ++i) {
            const uint64_t n = pc_off[i + 1] - pc_off[i], need = unit_tok_cap(n);
            if (used && used + need > cap_tokens) {
                p->rounds.push_back({first, i, round_max});
                first = i;
                used = 0;
                round_max = 0;
            }
            if (n > round_max) round_max = n;
            tok_off[i] = used;
            used += need;
            if (used > tok_buf_tokens) tok_buf_tokens = used;
            if (n_blocks + unit_blocks(n) > 0xffffffffull) return ZGPU_E_TOOBIG;
            blk_first[i] = (uint32_t)n_blocks;
            n_blocks += unit_blocks(n);
        }
        p->rounds.push_back({first, n_pieces, round_max});
    }
    // ---- size classes: a round whose units differ in class is searched by one launch per class, each over
    // a list of unit numbers (the variants with smaller windows run more workgroups per CU, and a unit that
    // fits the smaller window loses nothing to it)
    std::vector<uint32_t> lists;
    p->search_by_class = kHalfWindow < 0 && p->knobs.nice <= 32;
    if (ctype != ZGPU_NONE) {
        auto cls_of = [](uint64_t n) { return n <= kQuarterMax ? 0 : n <= kHalfWindowMax ? 1 : 2; };
        for (auto &r : p->rounds) {
            uint32_t cnt[3] = {0, 0, 0};
            for (uint32_t i = r.first; i < r.end; ++i) ++cnt[cls_of(pc_off[i + 1] - pc_off[i])];
            if ((cnt[0] != 0) + (cnt[1] != 0) + (cnt[2] != 0) < 2) continue;
            for (int c = 0; c < 3; ++c) {
                r.cls_at[c] = (uint32_t)lists.size();
                r.cls_len[c] = cnt[c];
                for (uint32_t i = r.first; i < r.end; ++i)
                    if (cls_of(pc_off[i + 1] - pc_off[i]) == c) lists.push_back(i);
            }
        }
    }
    // Units go straight to their final offsets (sizes first, a look-back for the offsets) unless much of the
    // plan is small units: thousands of those are in flight at once, and a look-back per
    // unit then costs more than staging the small output in slots and gathering it.
    // Decided by bytes: a few small files beside large ones ride along in the direct path, a large file
    // among many small ones is staged like them.
    {
        uint64_t small_bytes = 0;
        for (uint32_t i = 0; i < n_pieces; ++i) {
            const uint64_t n = pc_off[i + 1] - pc_off[i];
            if (n <= kEmitSmallMax) small_bytes += n ? n : 1;
        }
        p->direct = ctype != ZGPU_NONE && small_bytes * 8 <= pc_off[n_pieces] - pc_off[0];
        if (small_bytes == 0) p->direct = ctype != ZGPU_NONE;
    }
    // ---- checksum segments (only when a unit is longer than a segment) ----
    std::vector<uint64_t> seg_off;
    std::vector<uint32_t> first_seg;
    p->n_segs = 0;
    if (flags & (ZGPU_F_WANT_CRC | ZGPU_F_WANT_ADLER)) {
        bool long_unit = false;
        for (uint32_t i = 0; i < n_units && !long_unit; ++i) long_unit = in_off[i + 1] - in_off[i] > ZGPU_CK_SEG;
        if (long_unit) {
            for (uint32_t i = 0; i < n_units; ++i) {
                first_seg.push_back((uint32_t)seg_off.size());
                const uint64_t o = in_off[i], n = in_off[i + 1] - o;
                uint64_t k = 0;
                do {
                    seg_off.push_back(o + k);
                    k += ZGPU_CK_SEG;
                } while (k < n);
                if (seg_off.size() > 0x7fffffffull) return ZGPU_E_TOOBIG;
            }
            first_seg.push_back((uint32_t)seg_off.size());
            p->n_segs = (uint32_t)seg_off.size();
            seg_off.push_back(in_off[n_units]);
        }
    }
    // ---- one blob for the tables the kernels read ----
    size_t off = 0;
    auto place = [&](size_t bytes) {
        const size_t at = off;
        off = align_up(off + bytes, 16);
        return at;
    };
    const size_t o_in = place((size_t)(n_pieces + 1) * 8), o_slot = place((size_t)(n_pieces + 1) * 8),
                 o_tok = place((size_t)n_pieces * 8), o_blk = place((size_t)n_pieces * 4),
                 o_prime = place((size_t)n_pieces * 4), o_final = place((size_t)n_pieces),
                 o_uin = place(p->split ? (size_t)(n_units + 1) * 8 : 0),
                 o_fp = place(p->split ? (size_t)(n_units + 1) * 4 : 0),
                 o_seg = place(p->n_segs ? (size_t)(p->n_segs + 1) * 8 : 0),
                 o_fseg = place(p->n_segs ? (size_t)(n_units + 1) * 4 : 0), o_list = place(lists.size() * 4);
    p->h_tab.resize(off);
    uint8_t *h = p->h_tab.data();
    memcpy(h + o_in, pc_off.data(), (size_t)(n_pieces + 1) * 8);
    memcpy(h + o_slot, slot.data(), (size_t)(n_pieces + 1) * 8);
    memcpy(h + o_tok, tok_off.data(), (size_t)n_pieces * 8);
    memcpy(h + o_blk, blk_first.data(), (size_t)n_pieces * 4);
    memcpy(h + o_prime, pc_prime.data(), (size_t)n_pieces * 4);
    memcpy(h + o_final, pc_final.data(), (size_t)n_pieces);
    if (p->split) {
        memcpy(h + o_uin, in_off, (size_t)(n_units + 1) * 8);
        memcpy(h + o_fp, first_piece.data(), (size_t)(n_units + 1) * 4);
    }
    if (!lists.empty()) memcpy(h + o_list, lists.data(), lists.size() * 4);
    if (p->n_segs) {
        memcpy(h + o_seg, seg_off.data(), (size_t)(p->n_segs + 1) * 8);
        memcpy(h + o_fseg, first_seg.data(), (size_t)(n_units + 1) * 4);
    }
    int rc;
#define ENS(buf, bytes)                        \
    if ((rc = (buf).ensure(bytes)) != ZGPU_OK) return rc
    ENS(p->d_tab, off);
    ENS(p->b_out_off, (size_t)(n_pieces + 1) * 8);
    ENS(p->b_out_size, (size_t)n_pieces * 4);
    ENS(p->b_status, (size_t)n_pieces * 4);
    ENS(p->b_crc, (size_t)n_units * 4);
    ENS(p->b_adler, (size_t)n_units * 4);
    ENS(p->b_counter, (p->rounds.size() * 6 + 2) * 4);
    if (p->direct) ENS(p->b_state, (size_t)n_pieces * 8);
    if (ctype != ZGPU_NONE && !p->direct) ENS(p->b_out_tmp, acc + 64);
    if (ctype != ZGPU_NONE) {
        ENS(p->b_tok, (size_t)tok_buf_tokens * 2 + 64);
        ENS(p->b_hist, (size_t)n_blocks * 320 * 4);
        ENS(p->b_ntok, (size_t)n_blocks * 4);
    }
    // the dense-class pass: not where the whole plan is searched thoroughly or at least as deep anyway
    p->dense_cap = 0;
    if (ctype != ZGPU_NONE && p->knobs.dense_chain > p->knobs.max_chain && p->knobs.nice <= 32) {
        for (const auto &r : p->rounds)
            if (r.end - r.first > p->dense_cap) p->dense_cap = r.end - r.first;
        ENS(p->b_dense, (size_t)p->dense_cap * 4);
        ENS(p->b_taken, (size_t)n_pieces);
        // (the plans of the host-buffer path keep both searches on their slot's stream: its sub-batches already
        // overlap each other's kernels, and four more streams on the process's four hardware queues put a
        // slot's copies behind another slot's kernels -- Node, 1 GiB page-locked in and out: 29 -> 40 ms)
        if (!p->pooled) {
            if (!p->side) HIPCHK(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
            if (!p->ev_fork) HIPCHK(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
            if (!p->ev_join) HIPCHK(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
        }
    }
    if (p->split) ENS(p->b_unit_out_off, (size_t)(n_units + 1) * 8);
    if (p->n_segs) {
        ENS(p->b_seg_crc, (size_t)p->n_segs * 4);
        ENS(p->b_seg_adler, (size_t)p->n_segs * 4);
    }
#undef ENS
    uint8_t *d = p->d_tab.as<uint8_t>();
    p->d_in_off = (uint64_t *)(d + o_in);
    p->d_slot_off = (uint64_t *)(d + o_slot);
    p->d_tok_off = (uint64_t *)(d + o_tok);
    p->d_blk_first = (uint32_t *)(d + o_blk);
    p->d_prime = (uint32_t *)(d + o_prime);
    p->d_final = d + o_final;
    p->d_unit_in_off = (uint64_t *)(d + o_uin);
    p->d_first_piece = (uint32_t *)(d + o_fp);
    p->d_list = (uint32_t *)(d + o_list);
    p->d_seg_off = (uint64_t *)(d + o_seg);
    p->d_first_seg = (uint32_t *)(d + o_fseg);
    if (async) {
        // pageable source: the runtime has taken the bytes when the call returns
        HIPCHK(hipMemcpyAsync(d, h, off, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemsetAsync(p->b_crc.p, 0, (size_t)n_units * 4, st));
        HIPCHK(hipMemsetAsync(p->b_adler.p, 0, (size_t)n_units * 4, st));
    } else {
        HIPCHK(hipMemcpy(d, h, off, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(p->b_crc.p, 0, (size_t)n_units * 4));
        HIPCHK(hipMemset(p->b_adler.p, 0, (size_t)n_units * 4));
    }
    for (int i = 0; i < 5; ++i)
        if (!p->ev[i]) HIPCHK(hipEventCreate(&p->ev[i]));
    return ZGPU_OK;
}

// ---- host-buffer entry points: per-device staging kept from call to call ----------------
// A batch in host memory is cut into sub-batches of at most kSubBytes of input (64 MiB).  Four slots per
// device (stream, input buffer, plan with its workspace) take them in turn, each driven by its
// own host thread: the upload of one sub-batch runs while another is compressed and a third is
// fetched, so PCIe and the kernels overlap.  Nothing here is allocated per call once the
// buffers have grown to the caller's batch size.
constexpr int kMaxSlots = 8;
// tuning overrides (read once): ZGPU_SUB_BYTES, ZGPU_SLOTS
const uint64_t kSubBytes = [] {
    const char *e = getenv("ZGPU_SUB_BYTES");
    const uint64_t v = e ? strtoull(e, nullptr, 10) : 0;
    return v >= (1ull << 20) ? v : (64ull << 20);
}();
constexpr uint64_t kStageMaxBytes = 3ull << 30;
// the page-locked staging buffer a device keeps between two-step calls is at most this large
// (ZGPU_SPARE_HOST_BYTES overrides; larger ones are released when their job ends)
const uint64_t kSpareHostBytes = [] {
    const char *e = getenv("ZGPU_SPARE_HOST_BYTES");
    return e ? strtoull(e, nullptr, 10) : (1536ull << 20);
}();
const int kSlots = [] {
    const char *e = getenv("ZGPU_SLOTS");
    const int v = e ? atoi(e) : 0;
    return v >= 2 && v <= kMaxSlots ? v : 4;
}();
const int kComputeGate = [] {
    const char *e = getenv("ZGPU_COMPUTE_GATE");
    const int v = e ? atoi(e) : 0;
    return v >= 1 && v <= kMaxSlots ? v : 2;
}();
struct Slot {
    hipStream_t st = nullptr;
    DevBuf d_in;
    zgpu_plan *plan = nullptr;
    DevBuf ck_off, ck_crc, ck_adler;  // segment tables of the host checksum call
};
// a counting gate for the stages of the host pipeline
struct Gate {
    std::mutex mu;
    std::condition_variable cv;
    int free_;
    explicit Gate(int n) : free_(n) {}
    void enter() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return free_ > 0; });
        --free_;
    }
    void leave() {
        {
            std::lock_guard<std::mutex> lk(mu);
            ++free_;
        }
        cv.notify_one();
    }
};
struct GateHold {
    Gate *g;
    explicit GateHold(Gate &g_, bool use = true) : g(use ? &g_ : nullptr) {
        if (g) g->enter();
    }
    ~GateHold() {
        if (g) g->leave();
    }
};
// whether a host pointer lies in page-locked memory (hipHostMalloc / hipHostRegister, zgpu_host_alloc)
bool is_page_locked(const void *p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // ordinary memory is reported as an error: not one of ours
        return false;
    }
    return a.type == hipMemoryTypeHost;
}
struct HostCtx {
    std::mutex mu;  // one host-buffer call at a time per device
    // The slots' threads take turns at the three resources a sub-batch needs one after the other: the link to the
    // device, the compute units, the link back.  Left to themselves the threads fall into step -- all upload at
    // once, each at a quarter of the rate, then all compute, then all download: the phases follow each other
    // instead of overlapping (a trace of 1 GiB: 22 ms of kernels inside a 40 ms call).  One upload and one
    // download at a time, and two sub-batches' kernels (the second fills the compute units the first leaves idle
    // at its end), make it a pipeline.
    // (Only for page-locked buffers, which move by DMA at the link's rate: from ordinary memory the runtime
    // copies through its own staging on the calling thread, and there several threads side by side are what
    // reaches the link's rate -- those calls pass the upload / download gates without waiting.)
    Gate up{1}, compute{kComputeGate}, down{1};
    Slot slot[kMaxSlots];
    DevBuf d_res;   // compressed bytes of a batch until they are copied out (spare; a job takes it)
    HostBuf h_res;  // ... and their page-locked landing place on the host for two-step calls
    DevBuf ck_res;
};

void free_host_ctx(HostCtx *c) {
    if (!c) return;
    for (Slot &s : c->slot) {
        if (s.plan) zgpu_plan_destroy(s.plan);
        s.d_in.release();
        s.ck_off.release();
        s.ck_crc.release();
        s.ck_adler.release();
        if (s.st) (void)hipStreamDestroy(s.st);
    }
    c->d_res.release();
    c->h_res.release();
    c->ck_res.release();
    delete c;
}

// the context of the current device (created on first use)
int get_host_ctx(DeviceState *ds, HostCtx **out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!ds->host) {
        auto *c = new (std::nothrow) HostCtx();
        if (!c) return ZGPU_E_NOMEM;
        for (int si = 0; si < kSlots; ++si) {
            Slot &s = c->slot[si];
            hipError_t e = hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking);
            if (e == hipSuccess) {
                s.plan = new (std::nothrow) zgpu_plan();
                if (s.plan) {
                    s.plan->ds = ds;
                    s.plan->pooled = true;
                }
            }
            if (e != hipSuccess || !s.plan) {
                g_last_hip = (int)e;
                free_host_ctx(c);
                return e != hipSuccess ? ZGPU_E_NODEVICE : ZGPU_E_NOMEM;
            }
        }
        ds->host = c;
    }
    *out = ds->host;
    return ZGPU_OK;
}

// Contiguous ranges of units for n_dev devices, balanced by input bytes: range d ends at the
// first unit boundary at or after d + 1 parts of the total (units are never cut here).
void split_units_by_bytes(const uint64_t *in_off, uint32_t n_units, int n_dev, std::vector<uint32_t> &cut) {
    cut.assign((size_t)n_dev + 1, n_units);
    cut[0] = 0;
    const uint64_t base = in_off[0], total = in_off[n_units] - base;
    uint32_t u = 0;
    for (int d = 1; d < n_dev; ++d) {
        const uint64_t want = base + (total / (uint64_t)n_dev) * (uint64_t)d + (total % (uint64_t)n_dev) * (uint64_t)d / (uint64_t)n_dev;
        while (u < n_units && in_off[u] < want) ++u;
        cut[d] = u;
    }
}

struct SubBatch {
    uint32_t first = 0, end = 0;  // units
    int dev_index = 0;            // which of the job's devices
    uint64_t res_off = 0;         // where its compressed bytes start in that device's result buffer
    uint64_t res_cap = 0;         // and how many that region holds
    uint64_t out_bytes = 0;       // their size, once compressed
};
}  // namespace

// a host-buffer batch between its two calls: sizes known, compressed bytes still on the device(s)
struct zgpu_job {
    struct Dev {
        int device = -1;
        DeviceState *ds = nullptr;
        HostCtx *ctx = nullptr;
        DevBuf d_res;
        HostBuf h_res;       // two-step calls: the compressed bytes come down while later sub-batches run
        bool staged = false;
        uint64_t res_need = 0;
    };
    std::vector<Dev> devs;
    std::vector<SubBatch> subs;
    std::vector<uint64_t> out_off;  // per unit, n + 1
    std::vector<uint32_t> crc, adler;
    uint32_t n_units = 0;
};

namespace {

// compresses sub-batch `sb` of the job on slot `sl` of its device; when out != null and the
// sub-batches before it are placed, also copies its bytes to their final place in out
struct BatchShared {
    const uint8_t *in;
    const uint64_t *in_off;
    uint32_t n_units;
    int ctype, lazy;
    uint32_t flags;
    zgpu_job *job;
    // ordered placement (one device, caller's buffer known): sub-batch i may copy out once the
    // sizes of all sub-batches before it are known
    uint8_t *out = nullptr;
    uint64_t out_cap = 0;
    bool in_locked = false, out_locked = false;  // the caller's buffers are page-locked
    std::mutex mu;
    std::condition_variable cv;
    size_t placed = 0;       // sub-batches [0, placed) have their sizes in
    uint64_t placed_bytes = 0;
    int rc = ZGPU_OK;        // first error
};

int run_sub(BatchShared &S, size_t si, Slot &sl) {
    zgpu_job *j = S.job;
    SubBatch &sb = j->subs[si];
    zgpu_job::Dev &dv = j->devs[sb.dev_index];
    const uint32_t n = sb.end - sb.first;
    const uint64_t *go = S.in_off + sb.first;
    const bool chunked = (S.flags & ZGPU_F_SYNC_ALIGN) && !(S.flags & ZGPU_F_FINAL_EACH);
    // bytes of the same stream right before the sub-batch's first unit travel with it
    uint64_t hist = 0;
    if (chunked && (S.flags & ZGPU_F_PRIME_WINDOW)) {
        hist = go[0] - S.in_off[0];
        if (S.flags & ZGPU_F_PRIME_FIRST) hist = go[0];  // the stream began before the batch
        if (hist > 32768) hist = 32768;
    }
    const uint64_t lo = go[0] - hist, hi = go[n];
    const uint64_t lead = lo & 255u;  // the kernels see the caller's alignment (modulo 256)
    // unit offsets relative to the slot's input buffer
    std::vector<uint64_t> rel((size_t)n + 1);
    for (uint32_t i = 0; i <= n; ++i) rel[i] = go[i] - lo + lead;
    uint32_t flags = S.flags;
    if (sb.end != S.n_units) flags &= ~(uint32_t)ZGPU_F_FINAL_LAST;
    int rc = ZGPU_OK;
    if ((rc = sl.d_in.ensure(lead + (hi - lo) + 64)) != ZGPU_OK) return rc;
    {
        GateHold up(dv.ctx->up, S.in_locked);
        rc = plan_setup(sl.plan, rel.data(), n, S.ctype, S.lazy, flags, hist, sl.st, true);
        if (rc) return rc;
        if (hi > lo) HIPCHK(hipMemcpyAsync(sl.d_in.as<uint8_t>() + lead, S.in + lo, hi - lo, hipMemcpyHostToDevice, sl.st));
        HIPCHK(hipStreamSynchronize(sl.st));
    }
    uint8_t *d_out = dv.d_res.as<uint8_t>() + sb.res_off;
    // offsets relative to this sub-batch's result region; the n + 1 entries go through a local
    // array because entry n would land on the next sub-batch's first
    std::vector<uint64_t> loc((size_t)n + 1);
    {
        GateHold compute(dv.ctx->compute);
        rc = zgpu_plan_run(sl.plan, sl.d_in.p, d_out, sb.res_cap, sl.st);
        if (rc) return rc;
        rc = zgpu_plan_fetch(sl.plan, loc.data(), j->crc.empty() ? nullptr : j->crc.data() + sb.first,
                             j->adler.empty() ? nullptr : j->adler.data() + sb.first, sl.st);
        if (rc) return rc;
    }
    memcpy(j->out_off.data() + sb.first, loc.data(), (size_t)n * sizeof(uint64_t));
    sb.out_bytes = loc[n];
    if (dv.staged && sb.out_bytes) {
        GateHold down(dv.ctx->down);  // the library's own staging is page-locked
        HIPCHK(hipMemcpyAsync((uint8_t *)dv.h_res.p + sb.res_off, d_out, sb.out_bytes, hipMemcpyDeviceToHost, sl.st));
        HIPCHK(hipStreamSynchronize(sl.st));
    }
    if (!S.out) return ZGPU_OK;
    uint64_t at;
    {
        std::unique_lock<std::mutex> lk(S.mu);
        S.cv.wait(lk, [&] { return S.placed == si || S.rc != ZGPU_OK; });
        if (S.rc != ZGPU_OK) return ZGPU_OK;  // someone else failed; that error is reported
        at = S.placed_bytes;
        S.placed_bytes += sb.out_bytes;
        S.placed = si + 1;
    }
    S.cv.notify_all();
    if (at + sb.out_bytes > S.out_cap) return ZGPU_E_OUTCAP;
    if (sb.out_bytes) {
        GateHold down(dv.ctx->down, S.out_locked);
        HIPCHK(hipMemcpyAsync(S.out + at, d_out, sb.out_bytes, hipMemcpyDeviceToHost, sl.st));
        HIPCHK(hipStreamSynchronize(sl.st));
    }
    return ZGPU_OK;
}

void fail_shared(BatchShared &S, int rc) {
    {
        std::lock_guard<std::mutex> lk(S.mu);
        if (S.rc == ZGPU_OK) S.rc = rc;
    }
    S.cv.notify_all();
}

// The CPUs the system lists as local to a GPU (/sys/bus/pci/devices/<bus id>/local_cpulist), cut down to
// the ones this process may run on.  A device's host threads are bound to them: their copies to and from
// page-locked buffers and their launches then stay on the GPU's own socket.  Empty when the file cannot be
// read, names no CPU this process may use, or ZGPU_NO_AFFINITY=1.
#if defined(__linux__)
bool device_cpus(int device, cpu_set_t *out) {
    static const bool off = [] {
        const char *e = getenv("ZGPU_NO_AFFINITY");
        return e && e[0] == '1';
    }();
    if (off) return false;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return false;
    for (char *c = bus; *c; ++c) *c = (char)tolower((unsigned char)*c);
    char path[160];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bus);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    char line[4096] = {0};
    const bool got = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    CPU_ZERO(out);
    int n = 0;
    for (char *p = line; *p;) {  // "0-31,64-95"
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (c >= 0 && CPU_ISSET((int)c, &allowed)) {
                CPU_SET((int)c, out);
                ++n;
            }
        while (*p == ',' || *p == ' ' || *p == '\n') ++p;
    }
    return n > 0;
}
// binds the calling thread for the life of the object, and gives it back what it had
struct ThreadPlacement {
    cpu_set_t before;
    bool bound = false;
    explicit ThreadPlacement(int device) {
        cpu_set_t want;
        if (!device_cpus(device, &want)) return;
        if (pthread_getaffinity_np(pthread_self(), sizeof before, &before) != 0) return;
        bound = pthread_setaffinity_np(pthread_self(), sizeof want, &want) == 0;
    }
    ~ThreadPlacement() {
        if (bound) (void)pthread_setaffinity_np(pthread_self(), sizeof before, &before);
    }
};
#else
struct ThreadPlacement {
    explicit ThreadPlacement(int) {}
};
#endif

// all sub-batches of one device, slot threads side by side
void run_device(BatchShared &S, int dev_index) {
    zgpu_job::Dev &dv = S.job->devs[dev_index];
    std::vector<size_t> mine;
    for (size_t i = 0; i < S.job->subs.size(); ++i)
        if (S.job->subs[i].dev_index == dev_index) mine.push_back(i);
    if (mine.empty()) return;
    std::lock_guard<std::mutex> ctx_lock(dv.ctx->mu);
    // with several devices at work each one's threads stay on the CPUs next to it
    const bool place = S.job->devs.size() > 1;
    auto worker = [&](int slot) {
        if (hipSetDevice(dv.device) != hipSuccess) {
            fail_shared(S, ZGPU_E_NODEVICE);
            return;
        }
        std::unique_ptr<ThreadPlacement> placed(place ? new (std::nothrow) ThreadPlacement(dv.device) : nullptr);
        for (size_t k = (size_t)slot; k < mine.size(); k += kSlots) {
            {
                std::lock_guard<std::mutex> lk(S.mu);
                if (S.rc != ZGPU_OK) return;
            }
            const int rc = run_sub(S, mine[k], dv.ctx->slot[slot]);
            if (rc) {
                fail_shared(S, rc);
                return;
            }
        }
    };
    if (mine.size() == 1) {
        worker(0);  // the latency path: no thread is started for a small call
        return;
    }
    std::vector<std::thread> th;
    const int nth = (int)(mine.size() < (size_t)kSlots ? mine.size() : (size_t)kSlots);
    for (int s = 1; s < nth; ++s) th.emplace_back(worker, s);
    worker(0);
    for (auto &t : th) t.join();
}

}  // namespace

// ---- zgpu_checksum_device_noalloc: a checksum call that allocates nothing ------------------
// Segment offsets, per-segment results and the lookup tables of zgpu_checksum_kernel live in one
// file-scope device array, which every GPU holds from the moment the library is loaded.  In host
// code the array's name is only a handle: its device address is asked for once per device
// (hipGetSymbolAddress) and that pointer is what kernels and copies get.
namespace {
constexpr uint32_t kNaMaxSegs = 4096;                                  // 4 GiB in 1 MiB segments
constexpr size_t kNaTables = 0;                                        // u32[20 * 256 + 8]
constexpr size_t kNaSegOff = 21504;                                    // u64[kNaMaxSegs + 1]
constexpr size_t kNaCrc = kNaSegOff + ((kNaMaxSegs + 1) * 8 + 255) / 256 * 256;  // u32[kNaMaxSegs]
constexpr size_t kNaAdler = kNaCrc + kNaMaxSegs * 4;                   // u32[kNaMaxSegs]
constexpr size_t kNaBytes = kNaAdler + kNaMaxSegs * 4;
static_assert((20 * 256 + 8) * 4 <= kNaSegOff, "tables fit their slot");
}  // namespace
static __device__ __align__(256) unsigned char scratch[kNaBytes];

namespace {
constexpr int kMaxDevices = 64;
unsigned char *g_na_ptr[kMaxDevices] = {};  // device address of `scratch`, per device, once set up
std::mutex g_na_mu;
}  // namespace

// segment offsets 0, seg, 2 seg, ..., len
extern "C" __global__ void __launch_bounds__(256)
zgpu_na_offsets_kernel(uint64_t *seg_off, uint32_t n_segs, uint64_t seg, uint64_t len) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_segs) seg_off[i] = (uint64_t)i * seg;
    if (i == n_segs) seg_off[i] = len;
}

// CRC-32 and Adler-32 of the whole buffer from the per-segment values.  Both combines are sums
// over the segments: crc = xor_i crc_i * x^(8 * bytes after segment i) in GF(2)[x]/P, and
// s1 = 1 + sum_i (s1_i - 1), s2 = sum_i s2_i + sum_i (s1_i - 1) * (bytes after segment i) mod 65521.
// x^(8 * 2^b) mod P for b = 0..39, worked out by the compiler: x^(8 * bytes) is then one multiplication per set bit
// of the byte count instead of a square-and-multiply ladder per segment (the fold of 1024 segments of 4 MiB:
// 0.11 -> 0.02 ms)
constexpr uint32_t c_gf_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? 0xEDB88320u : 0u);
    }
    return p;
}
struct XpowTab {
    uint32_t v[40];
};
constexpr XpowTab make_xpow_tab() {
    XpowTab t{};
    uint32_t sq = 0x00800000u;  // x^8
    for (int b = 0; b < 40; ++b) {
        t.v[b] = sq;
        sq = c_gf_mul(sq, sq);
    }
    return t;
}
__device__ const XpowTab kXpowBytes = make_xpow_tab();
static_assert(make_xpow_tab().v[0] == 0x00800000u, "x^8");

extern "C" __global__ void __launch_bounds__(256)
zgpu_na_fold_kernel(const uint64_t *seg_off, const uint32_t *seg_crc, const uint32_t *seg_adler, uint32_t n_segs,
                    uint32_t *out2) {
    __shared__ uint32_t r_crc[4];
    __shared__ unsigned long long r_s1[4], r_s2[4];
    const uint32_t M = 65521u;
    const uint64_t len = seg_off[n_segs];
    uint32_t crc = 0;
    unsigned long long s1 = 0, s2 = 0;
    for (uint32_t i = threadIdx.x; i < n_segs; i += 256) {
        const uint64_t after = len - seg_off[i + 1];
        // x^(8 * after): the product of x^(8 * 2^b) over the set bits b of `after` (below 2^40: device memory)
        uint32_t r = 0x80000000u;
        for (uint64_t e = after; e; e &= e - 1) r = gf_mul(r, kXpowBytes.v[__builtin_ctzll(e)]);
        crc ^= gf_mul(seg_crc[i], r);
        const uint32_t a = seg_adler[i];
        const unsigned long long d1 = ((a & 0xffffu) + M - 1) % M;
        s1 += d1;
        s2 += (a >> 16) + d1 * (after % M) % M;
    }
    for (int d = 32; d > 0; d >>= 1) {
        crc ^= __shfl_down(crc, d, 64);
        s1 += __shfl_down(s1, d, 64);
        s2 += __shfl_down(s2, d, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        r_crc[wave] = crc;
        r_s1[wave] = s1;
        r_s2[wave] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out2[0] = r_crc[0] ^ r_crc[1] ^ r_crc[2] ^ r_crc[3];
        const unsigned long long t1 = (1 + r_s1[0] + r_s1[1] + r_s1[2] + r_s1[3]) % M;
        const unsigned long long t2 = (r_s2[0] + r_s2[1] + r_s2[2] + r_s2[3]) % M;
        out2[1] = (uint32_t)(t2 << 16 | t1);
    }
}

extern "C" {

int zgpu_init(const int *device_ids, int n_devices) {
    int count = 0;
    HIPCHK(hipGetDeviceCount(&count));
    if (count <= 0) return ZGPU_E_NODEVICE;
    if (n_devices < 0 || (n_devices > 0 && !device_ids)) return ZGPU_E_ARG;
    std::vector<int> ids;  // checked as a whole before it replaces the list in force
    for (int i = 0; i < n_devices; ++i) {
        if (device_ids[i] < 0 || device_ids[i] >= count) return ZGPU_E_ARG;
        ids.push_back(device_ids[i]);
    }
    std::lock_guard<std::mutex> lk(g_mu);
    g_allowed.swap(ids);
    return ZGPU_OK;
}

void *zgpu_host_alloc(uint64_t bytes) {
    void *p = nullptr;
    // portable: page-locked for every device, whichever is current now or later
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) {
        g_last_hip = (int)e;
        return nullptr;
    }
    return p;
}

void zgpu_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

void zgpu_host_trim(void) {
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    std::vector<DeviceState *> devs;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        devs = g_devs;
    }
    for (DeviceState *s : devs) {
        HostCtx *c;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            c = s->host;
        }
        if (!c) continue;
        std::lock_guard<std::mutex> ctx_lock(c->mu);  // no host-buffer call is running on this device
        DevBuf d;
        HostBuf h;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            std::swap(d, c->d_res);
            std::swap(h, c->h_res);
        }
        h.release();
        if (d.p && hipSetDevice(s->device) == hipSuccess) d.release();
    }
    if (have_cur) (void)hipSetDevice(cur);
}

int zgpu_device_count(void) {
    int count = 0;
    HIPCHK(hipGetDeviceCount(&count));
    return count;
}

void zgpu_shutdown(void) {
    if (g_live.load() > 0) return;  // plans or jobs still hold device state: destroy them first
    std::lock_guard<std::mutex> lk(g_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto *s : g_devs) {
        (void)hipSetDevice(s->device);
        free_state(s);
    }
    g_devs.clear();
    (void)hipSetDevice(cur);
}

const char *zgpu_strerror(int code) {
    switch (code) {
    case ZGPU_OK: return "ok";
    case ZGPU_E_NODEVICE: return "no usable HIP device or HIP runtime error";
    case ZGPU_E_ARG: return "invalid argument (e.g. invalid compression type)";
    case ZGPU_E_NOMEM: return "out of memory";
    case ZGPU_E_OUTCAP: return "output buffer too small";
    case ZGPU_E_UNIT: return "a unit overflowed its output slot";
    case ZGPU_E_TOOBIG: return "unit larger than ZGPU_MAX_UNIT_BYTES";
    default: return "unknown error";
    }
}

int zgpu_last_hip_error(void) { return g_last_hip; }

#ifdef ZGPU_PROFILE
// Diagnostic build only (not part of include/zgpu.h): per-phase cycle sums of
// thread 0 of every workgroup since the last reset.
// out: 32 sums -- 16 of the search kernel, then 16 of the emit kernel (thread 0 of every unit's group)
int zgpu_prof_read(unsigned long long *out32, int reset) {
    if (!g_prof) {
        HIPCHK(hipMalloc(&g_prof, 256));
        HIPCHK(hipMemset(g_prof, 0, 256));
    }
    HIPCHK(hipDeviceSynchronize());
    if (out32) HIPCHK(hipMemcpy(out32, g_prof, 256, hipMemcpyDeviceToHost));
    if (reset) HIPCHK(hipMemset(g_prof, 0, 256));
    return ZGPU_OK;
}
#endif

int zgpu_set_search(int max_chain, int nice_length) {
    if (max_chain < 1 || max_chain > 32768 || nice_length < 3 || nice_length > 258) return ZGPU_E_ARG;
    g_max_chain = max_chain;
    g_nice = nice_length;
    return ZGPU_OK;
}

int zgpu_set_long_match(int deep_chain, int probe_fans) {
    if (deep_chain < 0 || deep_chain > 32768 || probe_fans < 0 || probe_fans > 64) return ZGPU_E_ARG;
    g_deep = deep_chain;  // probe_fans: accepted, no longer used (see zgpu.h)
    return ZGPU_OK;
}

int zgpu_set_split(uint64_t piece_bytes) {
    if (piece_bytes && piece_bytes < (1ull << 20)) return ZGPU_E_ARG;
    g_split_bytes = piece_bytes;
    return ZGPU_OK;
}

int zgpu_set_dense_search(int max_chain) {
    if (max_chain < 0 || max_chain > 32768) return ZGPU_E_ARG;
    g_dense_chain = max_chain;
    return ZGPU_OK;
}

int zgpu_set_min_lazy(int min_lazy) {
    if (min_lazy < 0 || min_lazy > 258) return ZGPU_E_ARG;
    g_min_lazy = min_lazy;
    return ZGPU_OK;
}

uint64_t zgpu_deflate_bound(uint64_t in_len) {
    // worst-case literal codes, per-block headers, and for long units the framing of their pieces
    return in_len + in_len / 8 + (in_len / 65536 + 1) * 400 + 64 + (in_len >> 17) * 512;
}

int zgpu_plan_create(zgpu_plan **out, const uint64_t *in_off, uint32_t n_units, int ctype, int lazy,
                     uint32_t flags) {
    if (!out || !in_off || n_units == 0) return ZGPU_E_ARG;
    if (ctype < 0 || ctype > 2 || lazy < 0) return ZGPU_E_ARG;
    for (uint32_t i = 0; i < n_units; ++i) {  // argument errors come before any device work
        if (in_off[i + 1] < in_off[i]) return ZGPU_E_ARG;
        if (in_off[i + 1] - in_off[i] > ZGPU_MAX_UNIT_BYTES) return ZGPU_E_TOOBIG;
    }
    DeviceState *ds = nullptr;
    int rc = get_state(&ds);
    if (rc) return rc;
    auto *p = new (std::nothrow) zgpu_plan();
    if (!p) return ZGPU_E_NOMEM;
    ++g_live;
    p->ds = ds;
    uint64_t history = 0;
    if (flags & ZGPU_F_PRIME_FIRST) history = in_off[0] < 32768 ? in_off[0] : 32768;
    rc = plan_setup(p, in_off, n_units, ctype, lazy, flags, history, nullptr, false);
    if (rc) {
        zgpu_plan_destroy(p);
        return rc;
    }
    *out = p;
    return ZGPU_OK;
}

void zgpu_plan_destroy(zgpu_plan *p) {
    if (!p) return;
    p->release_buffers();
    for (int i = 0; i < 5; ++i)
        if (p->ev[i]) (void)hipEventDestroy(p->ev[i]);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    if (p->ev_join) (void)hipEventDestroy(p->ev_join);
    if (p->side) (void)hipStreamDestroy(p->side);
    if (!p->pooled) --g_live;
    delete p;
}

uint64_t zgpu_plan_out_bound(const zgpu_plan *p) { return p ? p->out_bound : 0; }

int zgpu_plan_run(zgpu_plan *p, const void *d_in, void *d_out, uint64_t out_cap, void *stream_) {
    if (!p || !d_out || (!d_in && p->in_total)) return ZGPU_E_ARG;
    if ((uintptr_t)d_out & 3) return ZGPU_E_ARG;  // units that share a 32-bit word with a neighbour are stitched bytewise
    hipStream_t st = (hipStream_t)stream_;
    HIPCHK(hipEventRecord(p->ev[0], st));
    HIPCHK(hipMemsetAsync(p->d_counter(), 0, (p->rounds.size() * 6 + 2) * 4, st));
    if (p->direct) HIPCHK(hipMemsetAsync(p->b_state.p, 0, (size_t)p->n_pieces * 8, st));
    HIPCHK(hipEventRecord(p->ev[1], st));
    if (p->ctype == ZGPU_NONE) {
        int grid = p->n_units < (uint32_t)p->ds->cus * 8 ? (int)p->n_units : p->ds->cus * 8;
        hipLaunchKernelGGL(zgpu_stored_kernel, dim3(grid), dim3(256), 0, st, (const uint8_t *)d_in,
                           p->d_in_off, p->d_slot_off, (uint8_t *)d_out, out_cap, p->d_out_off(), p->d_out_size(),
                           p->d_status(), p->n_units, p->flags);
        HIPCHK(hipEventRecord(p->ev[4], st));
    } else {
        for (size_t r = 0; r < p->rounds.size(); ++r) {
            const uint32_t first = p->rounds[r].first, end = p->rounds[r].end, cnt = end - first;
            Params P;
            P.in = (const uint8_t *)d_in;
            P.in_off = p->d_in_off;
            P.tok = p->b_tok.as<uint16_t>();
            P.tok_off = p->d_tok_off;
            P.blk_first = p->d_blk_first;
            P.hist = p->b_hist.as<uint32_t>();
            P.ntok = p->b_ntok.as<uint32_t>();
            // the caller's lazy is the threshold below which a match is deferred when the next
            // position matches longer; min_lazy (default 258: always) is its floor
            P.lazy = p->lazy > p->knobs.min_lazy ? p->lazy : p->knobs.min_lazy;
            P.max_chain = p->knobs.max_chain;
            P.nice = p->knobs.nice;
            P.deep = p->knobs.deep;
            P.unit_prime = p->d_prime;
            P.prof = g_prof;
            P.unit_class = nullptr;
            P.key3_max_bytes = p->ctype == ZGPU_FIXED || kKey3Dynamic ? kKey3Max : 0u;
            for (int c = 0; c < 3; ++c) P.class_chain[c] = P.class_lazy[c] = 0;
            // one launch over the round's range, or one per size class over its list
            const auto &R = p->rounds[r];
            if (p->dense_cap) {
                // which units are of the dense class (their list starts over with every round)
                HIPCHK(hipMemsetAsync(p->b_dense.p, 0xff, (size_t)p->dense_cap * 4, st));
                if (r) HIPCHK(hipMemsetAsync(p->d_counter() + 6 * p->rounds.size(), 0, 8, st));
                HIPCHK(hipMemsetAsync(p->b_taken.as<uint8_t>() + first, 0, cnt, st));
                hipLaunchKernelGGL(zgpu_classify_kernel, dim3((cnt + 3) / 4), dim3(256), 0, st, (const uint8_t *)d_in,
                                   p->d_in_off, first, end, p->b_dense.as<uint32_t>(),
                                   p->d_counter() + 6 * p->rounds.size() + 1, p->b_taken.as<uint8_t>(),
                                   p->ctype == ZGPU_FIXED ? 1u : 0u);
            }
            P.skip = p->dense_cap ? p->b_taken.as<uint8_t>() : nullptr;
            hipStream_t fs = st;  // the stream of the first search
            Params D;
            if (p->dense_cap) {
                // ... and those are searched as the reference searches (chain budget dense_chain, a match is good
                // enough only at 258) by one launch of the thorough variant over that list, on the caller's
                // stream, while the first search -- which passes them over (b_taken) -- runs on the plan's own
                // stream.  The second search is enqueued first so that its workgroups -- one per unit in flight,
                // each for milliseconds to tens of milliseconds -- are placed first, and the first search, whose
                // workgroups draw units from a queue, fills the other CUs and those that become free: a few dense
                // units then cost a batch nothing it can measure, where a second search after the first made
                // every batch wait for its slowest dense unit (two units of 35 ms in 2 GiB of Python packages:
                // 60 -> 107 ms one after the other, 60 -> 67 ms side by side).  The list ends at its first
                // 0xffffffff entry: a workgroup that draws it is done, so with no dense unit in the round the
                // launch takes a few microseconds.
                // Three classes, one byte per unit (b_taken).  (1) Units of at most six byte values are parsed as
                // the caller's lazy says (0: greedily, like the reference): a literal costs them eight or nine
                // bits and a byte of a match a fraction of one, so putting a match off for a longer one loses
                // (two values: 1.04 x the reference's size under FIXED with the look-ahead, 1.00 x without;
                // 16-bit samples below six: 1.06 x against 1.00 x).  (2, 3) The others keep the look-ahead of the
                // default parse under FIXED, where a literal has its fixed price (tables of small numbers among
                // zeros: 1.05 x without, 1.03 x with); under DYNAMIC the code lengths follow the tokens and the
                // reference's sizes come from its greedy parse of the same matches (16-bit tables in shared
                // libraries: 1.03-1.05 x with the look-ahead, within the bar without).  (2) is short runs of
                // seven or eight values, which need the whole chain budget (1.05 x at an eighth of it).  (3) is
                // what else the statistics take: units of many byte values of which a few weigh heavily -- half
                // of the units of a shared library, a tenth of indented source text.  The whole budget costs a
                // batch of such units 18 x its time for 4.4 % of size under FIXED and 0.6 % under DYNAMIC; a
                // thirty-second of it (64 candidates, each compared to its full length) costs 1.5 x and buys
                // 3.0 % and 0.3 %, which is what they get.
                const int floor_lazy = p->lazy > p->knobs.min_lazy ? p->lazy : p->knobs.min_lazy;
                const int rest_lazy = p->ctype == ZGPU_FIXED ? floor_lazy : p->lazy;
                int short_chain = (p->knobs.dense_chain + 31) / 32;
                if (short_chain <= p->knobs.max_chain) short_chain = p->knobs.max_chain + 1;
                if (p->side) {
                    HIPCHK(hipEventRecord(p->ev_fork, st));
                    HIPCHK(hipStreamWaitEvent(p->side, p->ev_fork, 0));
                    fs = p->side;
                }
                D = P;
                D.skip = nullptr;
                D.unit_list = p->b_dense.as<uint32_t>();
                D.list_len = cnt;
                D.unit_first = 0;
                D.unit_end = 0xffffffffu;
                D.counter = p->d_counter() + 6 * p->rounds.size();
                D.unit_class = p->b_taken.as<uint8_t>();
                D.class_chain[0] = D.class_chain[1] = D.max_chain = p->knobs.dense_chain;
                D.class_chain[2] = short_chain;
                D.class_lazy[0] = D.lazy = p->lazy;
                D.class_lazy[1] = D.class_lazy[2] = rest_lazy;
                D.nice = 258;
                if (p->side) hipLaunchKernelGGL(zgpu_deflate_thorough_kernel, dim3(cnt < (uint32_t)p->ds->cus ? cnt : (uint32_t)p->ds->cus), dim3(1024), 0, st, D);
            }
            const bool has_lists = R.cls_len[0] + R.cls_len[1] + R.cls_len[2] != 0;
            const bool listed = has_lists && p->search_by_class;
            for (int c = 0; c < 3; ++c) {
                uint32_t n_here = cnt;
                int cls = c;
                P.counter = p->d_counter() + 6 * r + c;
                if (listed) {
                    if (!R.cls_len[c]) continue;
                    n_here = R.cls_len[c];
                    P.unit_list = p->d_list + R.cls_at[c];
                    P.list_len = n_here;
                    P.unit_first = 0;
                    P.unit_end = 0xffffffffu;
                } else {
                    if (c) break;
                    P.unit_list = nullptr;
                    P.list_len = 0;
                    P.unit_first = first;
                    P.unit_end = end;
                    // ZGPU_HALF_WINDOW=1 forces the half-window variant for any unit size, =0 the full window
                    cls = kHalfWindow == 1 ? 1 : (kHalfWindow == 0 || has_lists) ? 2 : R.max_bytes <= kQuarterMax ? 0 : R.max_bytes <= kHalfWindowMax ? 1 : 2;
                }
                const uint32_t cus = (uint32_t)p->ds->cus;
                if (p->knobs.nice > 32) {
                    hipLaunchKernelGGL(zgpu_deflate_thorough_kernel, dim3(n_here < cus ? n_here : cus), dim3(1024), 0, fs, P);
                } else if (cls == 0) {
                    // tiny units: a unit's few items keep eight waves busy as well as sixteen, and four units
                    // per CU are in flight instead of two
                    hipLaunchKernelGGL(zgpu_deflate_quarter_kernel, dim3(n_here < 4 * cus ? n_here : 4 * cus), dim3(512), 0, fs, P);
                } else if (cls == 1) {
                    // units that fit half a window lose nothing to it: two workgroups per CU, 32 waves
                    hipLaunchKernelGGL(zgpu_deflate_half_kernel, dim3(n_here < 2 * cus ? n_here : 2 * cus), dim3(1024), 0, fs, P);
                } else
                    hipLaunchKernelGGL(zgpu_deflate_kernel, dim3(n_here < cus ? n_here : cus), dim3(1024), 0, fs, P);
            }
            // the second search of the classified units has run beside this one: the emit kernel needs both
            // (one stream: the second search follows the first.  In front of it, its workgroups -- a CU's worth of
            // LDS each, though they have nothing to do -- would wait for the CUs that the sub-batch before this
            // one still holds, and this sub-batch's first search behind them could no longer fill the CUs as
            // they become free: Node, 1 GiB page-locked in and out, 29 -> 33 ms)
            if (p->dense_cap && !p->side)
                hipLaunchKernelGGL(zgpu_deflate_thorough_kernel, dim3(cnt < (uint32_t)p->ds->cus ? cnt : (uint32_t)p->ds->cus), dim3(1024), 0, st, D);
            if (fs != st) {
                HIPCHK(hipEventRecord(p->ev_join, fs));
                HIPCHK(hipStreamWaitEvent(st, p->ev_join, 0));
            }
            if (r == 0) HIPCHK(hipEventRecord(p->ev[4], st));
            EParams E;
            E.in_off = p->d_in_off;
            E.out = p->direct ? (uint8_t *)d_out : p->b_out_tmp.as<uint8_t>();
            E.out_cap = out_cap;
            E.slot_off = p->d_slot_off;
            E.out_off = p->d_out_off();
            E.state = p->b_state.as<unsigned long long>();
            E.n_total = p->n_pieces;
            E.out_size = p->d_out_size();
            E.status = p->d_status();
            E.tok = p->b_tok.as<uint16_t>();
            E.tok_off = p->d_tok_off;
            E.blk_first = p->d_blk_first;
            E.hist = p->b_hist.as<uint32_t>();
            E.ntok = p->b_ntok.as<uint32_t>();
            E.unit_final = p->d_final;
            E.ctype = p->ctype;
            E.prof = g_prof ? g_prof + 16 : nullptr;
            // one launch over the round's range; a staged round with units of several size classes is
            // emitted per class: the small ones a wave each, the others a workgroup each
            const bool by_class = has_lists && !p->direct && R.cls_len[2] != 0;  // all small: one launch does
            for (int c = 0; c < 3; ++c) {
                uint32_t n_here = cnt;
                bool small = R.max_bytes <= kEmitSmallMax;
                E.counter = p->d_counter() + 6 * r + 3 + c;
                if (by_class) {
                    if (!R.cls_len[c]) continue;
                    n_here = R.cls_len[c];
                    small = c < 2;
                    E.unit_list = p->d_list + R.cls_at[c];
                    E.list_len = n_here;
                    E.unit_first = 0;
                    E.unit_end = 0xffffffffu;
                } else {
                    if (c) break;
                    E.unit_list = nullptr;
                    E.list_len = 0;
                    E.unit_first = first;
                    E.unit_end = end;
                }
                if (small && !p->direct) {
                    // small units: one wave per unit, four units per workgroup, four workgroups per CU
                    const uint32_t wgs = (n_here + 3) / 4, emax = (uint32_t)p->ds->cus * 4;
                    hipLaunchKernelGGL(zgpu_emit_small_kernel, dim3(wgs < emax ? wgs : emax), dim3(256), 0, st, E);
                } else {
                    const uint32_t emax = (uint32_t)p->ds->cus * ZGPU_EMIT_WG_PER_CU;
                    if (p->direct)
                        hipLaunchKernelGGL(zgpu_emit_kernel, dim3(n_here < emax ? n_here : emax), dim3(256), 0, st, E);
                    else
                        hipLaunchKernelGGL(zgpu_emit_slots_kernel, dim3(n_here < emax ? n_here : emax), dim3(256), 0, st, E);
                }
            }
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(p->ev[2], st));
    if (p->ctype != ZGPU_NONE && !p->direct) {
        hipLaunchKernelGGL(zgpu_scan_kernel, dim3(1), dim3(1024), 0, st, p->d_out_size(), p->d_out_off(), p->n_pieces);
        int grid = p->n_pieces < (uint32_t)p->ds->cus * 8 ? (int)p->n_pieces : p->ds->cus * 8;
        hipLaunchKernelGGL(zgpu_compact_kernel, dim3(grid), dim3(256), 0, st, p->b_out_tmp.as<uint8_t>(), p->d_slot_off,
                           p->d_out_off(), (uint8_t *)d_out, out_cap, p->d_status(), p->n_pieces);
    }
    if (p->split)
        hipLaunchKernelGGL(zgpu_unit_offsets_kernel, dim3(p->n_units / 256 + 1), dim3(256), 0, st, p->d_out_off(),
                           p->d_first_piece, p->d_unit_out_off(), p->n_units);
    if (p->flags & (ZGPU_F_WANT_CRC | ZGPU_F_WANT_ADLER)) {
        if (p->n_segs) {
            // long units: segments over many workgroups, then one fold per unit
            int grid = p->n_segs < (uint32_t)p->ds->cus * 8 ? (int)p->n_segs : p->ds->cus * 8;
            hipLaunchKernelGGL(zgpu_checksum_kernel, dim3(grid), dim3(ZGPU_CK_THREADS), 0, st, (const uint8_t *)d_in,
                               p->d_seg_off, p->n_segs, p->ds->d_tables, p->b_seg_crc.as<uint32_t>(),
                               p->b_seg_adler.as<uint32_t>());
            hipLaunchKernelGGL(zgpu_unit_fold_kernel, dim3(p->n_units / 256 + 1), dim3(256), 0, st, p->d_seg_off,
                               p->d_first_seg, p->b_seg_crc.as<uint32_t>(), p->b_seg_adler.as<uint32_t>(), p->n_units,
                               p->d_crc(), p->d_adler());
        } else if (p->n_units >= 1024 && p->in_total / p->n_units <= (32u << 10)) {
            // many small units: a wave each, four side by side in a workgroup (their time is the chain of
            // round trips and the fold of the lanes' remainders, not their bytes)
            const uint32_t wgs = (p->n_units + 3) / 4, gmax = (uint32_t)p->ds->cus * 6;
            hipLaunchKernelGGL(zgpu_checksum_wave_kernel, dim3(wgs < gmax ? wgs : gmax), dim3(ZGPU_CK_THREADS), 0, st,
                               (const uint8_t *)d_in, p->split ? p->d_unit_in_off : p->d_in_off, p->n_units,
                               p->ds->d_tables, p->d_crc(), p->d_adler());
        } else {
            int grid = p->n_units < (uint32_t)p->ds->cus * 8 ? (int)p->n_units : p->ds->cus * 8;
            hipLaunchKernelGGL(zgpu_checksum_kernel, dim3(grid), dim3(ZGPU_CK_THREADS), 0, st,
                               (const uint8_t *)d_in, p->split ? p->d_unit_in_off : p->d_in_off, p->n_units,
                               p->ds->d_tables, p->d_crc(), p->d_adler());
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(p->ev[3], st));
    p->ran = true;
    return ZGPU_OK;
}

const uint64_t *zgpu_plan_out_off_device(const zgpu_plan *p) {
    return p ? (p->split ? p->d_unit_out_off() : p->d_out_off()) : nullptr;
}
const uint32_t *zgpu_plan_crc_device(const zgpu_plan *p) { return p ? p->d_crc() : nullptr; }
const uint32_t *zgpu_plan_adler_device(const zgpu_plan *p) { return p ? p->d_adler() : nullptr; }

int zgpu_plan_fetch(zgpu_plan *p, uint64_t *out_off, uint32_t *crc32, uint32_t *adler32, void *stream_) {
    if (!p || !p->ran) return ZGPU_E_ARG;
    hipStream_t st = (hipStream_t)stream_;
    try {
        p->h_status.resize(p->n_pieces);
    } catch (const std::bad_alloc &) {
        return ZGPU_E_NOMEM;
    }
    // the copies ride on the run's stream (not on the null stream, which would wait for every other
    // stream of the process); one wait at the end
    HIPCHK(hipMemcpyAsync(p->h_status.data(), p->d_status(), (size_t)p->n_pieces * 4, hipMemcpyDeviceToHost, st));
    if (out_off)
        HIPCHK(hipMemcpyAsync(out_off, p->split ? p->d_unit_out_off() : p->d_out_off(), (size_t)(p->n_units + 1) * 8,
                              hipMemcpyDeviceToHost, st));
    if (crc32) HIPCHK(hipMemcpyAsync(crc32, p->d_crc(), (size_t)p->n_units * 4, hipMemcpyDeviceToHost, st));
    if (adler32) HIPCHK(hipMemcpyAsync(adler32, p->d_adler(), (size_t)p->n_units * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int rc = ZGPU_OK;
    for (uint32_t i = 0; i < p->n_pieces; ++i) {
        if (p->h_status[i] == 2) rc = ZGPU_E_OUTCAP;
        else if (p->h_status[i] && rc == ZGPU_OK) rc = ZGPU_E_UNIT;
    }
    return rc;
}

int zgpu_plan_kernel_ms(zgpu_plan *p, float *deflate_ms, float *total_ms) {
    if (!p || !p->ran) return ZGPU_E_ARG;
    HIPCHK(hipEventSynchronize(p->ev[3]));
    if (deflate_ms) HIPCHK(hipEventElapsedTime(deflate_ms, p->ev[1], p->ev[2]));
    if (total_ms) HIPCHK(hipEventElapsedTime(total_ms, p->ev[0], p->ev[3]));
    return ZGPU_OK;
}

int zgpu_plan_phase_ms(zgpu_plan *p, float *search_ms, float *emit_ms) {
    if (!p || !p->ran) return ZGPU_E_ARG;
    HIPCHK(hipEventSynchronize(p->ev[3]));
    if (search_ms) HIPCHK(hipEventElapsedTime(search_ms, p->ev[1], p->ev[4]));
    if (emit_ms) {
        *emit_ms = 0.f;
        if (p->rounds.size() == 1) HIPCHK(hipEventElapsedTime(emit_ms, p->ev[4], p->ev[2]));
    }
    return ZGPU_OK;
}

// ---- host-buffer batches ------------------------------------------------------------------
int zgpu_host_devices(int *device_ids, int cap) {
    std::vector<int> devs;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        devs = g_allowed;
    }
    if (devs.empty()) {
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) return ZGPU_E_NODEVICE;
        devs.push_back(cur);
    }
    for (int i = 0; i < (int)devs.size() && i < cap; ++i)
        if (device_ids) device_ids[i] = devs[i];
    return (int)devs.size();
}

int zgpu_partition_units(const uint64_t *in_off, uint32_t n_units, int n_parts, uint32_t *cut) {
    if (!in_off || !cut || n_parts < 1) return ZGPU_E_ARG;
    for (uint32_t i = 0; i < n_units; ++i)
        if (in_off[i + 1] < in_off[i]) return ZGPU_E_ARG;
    std::vector<uint32_t> c;
    split_units_by_bytes(in_off, n_units, n_parts, c);
    memcpy(cut, c.data(), ((size_t)n_parts + 1) * sizeof(uint32_t));
    return ZGPU_OK;
}

static int host_batch_compute(const uint8_t *in, const uint64_t *in_off, uint32_t n_units, int ctype, int lazy,
                              uint32_t flags, uint8_t *out, uint64_t out_cap, zgpu_job **job_out) {
    *job_out = nullptr;
    if (!in_off || n_units == 0) return ZGPU_E_ARG;
    if (ctype < 0 || ctype > 2 || lazy < 0) return ZGPU_E_ARG;
    for (uint32_t i = 0; i < n_units; ++i) {
        if (in_off[i + 1] < in_off[i]) return ZGPU_E_ARG;
        if (in_off[i + 1] - in_off[i] > ZGPU_MAX_UNIT_BYTES) return ZGPU_E_TOOBIG;
    }
    if (in_off[n_units] > in_off[0] && !in) return ZGPU_E_ARG;
    int prev_dev = 0;
    HIPCHK(hipGetDevice(&prev_dev));
    // devices: the ones zgpu_init named (all of them share a large batch), else the current one
    std::vector<int> devs;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        devs = g_allowed;
    }
    if (devs.empty()) devs.push_back(prev_dev);
    const uint64_t total = in_off[n_units] - in_off[0];
    // a device is worth its own range only from a few sub-batches of input on
    size_t use = (size_t)(total / (2 * kSubBytes)) + 1;
    if (use > devs.size()) use = devs.size();
    if (use > n_units) use = n_units;
    if (use == 1) {
        // small batch: the caller's current device if it is among the named ones
        for (int d : devs)
            if (d == prev_dev) devs[0] = prev_dev;
    }
    devs.resize(use);
    std::unique_ptr<zgpu_job> j(new (std::nothrow) zgpu_job());
    if (!j) return ZGPU_E_NOMEM;
    ++g_live;
    int rc = ZGPU_OK;
    auto restore = [&]() { (void)hipSetDevice(prev_dev); };
    try {
        j->n_units = n_units;
        j->out_off.assign((size_t)n_units + 1, 0);
        if (flags & ZGPU_F_WANT_CRC) j->crc.assign(n_units, 0);
        if (flags & ZGPU_F_WANT_ADLER) j->adler.assign(n_units, 0);
        std::vector<uint32_t> cut;
        split_units_by_bytes(in_off, n_units, (int)use, cut);
        j->devs.resize(use);
        for (size_t d = 0; d < use && rc == ZGPU_OK; ++d) {
            zgpu_job::Dev &dv = j->devs[d];
            dv.device = devs[d];
            if (hipSetDevice(dv.device) != hipSuccess) {
                rc = ZGPU_E_NODEVICE;
                break;
            }
            if ((rc = get_state(&dv.ds)) != ZGPU_OK) break;
            if ((rc = get_host_ctx(dv.ds, &dv.ctx)) != ZGPU_OK) break;
            // sub-batches of this device's range: about an eighth of the range each -- so that a call of moderate
            // size already overlaps its copies with its kernels -- between 8 MiB and kSubBytes, at least one unit
            const uint64_t range_bytes = in_off[cut[d + 1]] - in_off[cut[d]];
            uint64_t sub_bytes = range_bytes / 8;
            if (sub_bytes < (8ull << 20)) sub_bytes = 8ull << 20;
            if (sub_bytes > kSubBytes) sub_bytes = kSubBytes;
            uint32_t u = cut[d];
            while (u < cut[d + 1]) {
                SubBatch sb;
                sb.first = u;
                sb.dev_index = (int)d;
                uint64_t bytes = 0;
                do {
                    bytes += in_off[u + 1] - in_off[u];
                    ++u;
                } while (u < cut[d + 1] && bytes + (in_off[u + 1] - in_off[u]) <= sub_bytes);
                sb.end = u;
                sb.res_off = dv.res_need;
                uint64_t bound = 0;
                for (uint32_t i = sb.first; i < sb.end; ++i)
                    bound += align_up(zgpu_deflate_bound(in_off[i + 1] - in_off[i]) + 32, 16);
                // pieces of cut units add per-piece slack: the plan's own bound is what counts, and it
                // never exceeds the per-unit bound by more than the piece framing already inside it
                sb.res_cap = align_up(bound + bytes / 64 + (uint64_t)(sb.end - sb.first) * 64 + 4096, 256);
                dv.res_need += sb.res_cap;
                j->subs.push_back(sb);
            }
            // result buffer: the device's spare one if it is large enough, else a new one
            // a two-step call of moderate size also gets a page-locked host buffer: its sub-batches
            // come down as they finish, and the second step is a copy between host buffers
            const bool stage = !out && dv.res_need <= kStageMaxBytes;
            {
                std::lock_guard<std::mutex> lk(g_mu);
                std::swap(dv.d_res, dv.ctx->d_res);
                if (stage) std::swap(dv.h_res, dv.ctx->h_res);
            }
            rc = dv.d_res.ensure(dv.res_need + 64);
            if (rc == ZGPU_OK && stage) dv.staged = dv.h_res.ensure(dv.res_need + 64) == ZGPU_OK;
        }
    } catch (const std::bad_alloc &) {
        rc = ZGPU_E_NOMEM;
    }
    if (rc) {
        restore();
        zgpu_job_release(j.release());
        return rc;
    }
    BatchShared S;
    S.in = in;
    S.in_off = in_off;
    S.n_units = n_units;
    S.ctype = ctype;
    S.lazy = lazy;
    S.flags = flags;
    S.job = j.get();
    S.in_locked = is_page_locked(in);
    if (use == 1 && out) {  // one device and the caller's buffer at hand: copy out as sub-batches finish
        S.out = out;
        S.out_cap = out_cap;
        S.out_locked = is_page_locked(out);
    }
    if (use == 1) {
        run_device(S, 0);
    } else {
        std::vector<std::thread> th;
        for (size_t d = 1; d < use; ++d) th.emplace_back(run_device, std::ref(S), (int)d);
        run_device(S, 0);
        for (auto &t : th) t.join();
    }
    restore();
    if (S.rc) {
        zgpu_job_release(j.release());
        return S.rc;
    }
    // per-unit offsets: the sub-batches' own offsets shifted to their place in the whole output
    uint64_t at = 0;
    for (SubBatch &sb : j->subs) {
        // out_off[first..end) are relative to the sub-batch; out_off[end] is overwritten by the next
        // sub-batch's first entry (0), so its size was saved
        for (uint32_t i = sb.first; i < sb.end; ++i) j->out_off[i] += at;
        at += sb.out_bytes;
    }
    j->out_off[n_units] = at;
    *job_out = j.release();
    return ZGPU_OK;
}

// copies what is still on the devices to out (sub-batches side by side, two threads per device)
static int host_batch_copy_out(zgpu_job *j, uint8_t *out) {
    int prev_dev = 0;
    HIPCHK(hipGetDevice(&prev_dev));
    std::vector<uint64_t> at(j->subs.size());
    uint64_t a = 0;
    for (size_t i = 0; i < j->subs.size(); ++i) {
        at[i] = a;
        a += j->subs[i].out_bytes;
    }
    std::mutex mu;
    int rc = ZGPU_OK;
    bool all_staged = true;
    for (const zgpu_job::Dev &dv : j->devs) all_staged = all_staged && dv.staged;
    if (all_staged) {
        // everything already lies in page-locked host memory: plain copies, a few threads side by side
        // (the caller's buffer is often fresh, and first-touch page faults are what they spend time on)
        const int nth = a >= (64ull << 20) ? 4 : (a >= (4ull << 20) ? 2 : 1);
        auto mover = [&](int part) {
            for (size_t i = (size_t)part; i < j->subs.size(); i += (size_t)nth) {
                const SubBatch &sb = j->subs[i];
                if (sb.out_bytes)
                    memcpy(out + at[i], (const uint8_t *)j->devs[sb.dev_index].h_res.p + sb.res_off, sb.out_bytes);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nth; ++t) th.emplace_back(mover, t);
        mover(0);
        for (auto &t : th) t.join();
        return ZGPU_OK;
    }
    auto worker = [&](int dev_index, int part, int parts) {
        zgpu_job::Dev &dv = j->devs[dev_index];
        hipStream_t st = dv.ctx->slot[part].st;
        bool ok = hipSetDevice(dv.device) == hipSuccess;
        int k = 0;
        for (size_t i = 0; i < j->subs.size() && ok; ++i) {
            const SubBatch &sb = j->subs[i];
            if (sb.dev_index != dev_index || !sb.out_bytes) continue;
            if (k++ % parts != part) continue;
            ok = hipMemcpyAsync(out + at[i], dv.d_res.as<uint8_t>() + sb.res_off, sb.out_bytes, hipMemcpyDeviceToHost,
                                st) == hipSuccess;
        }
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        if (!ok) {
            std::lock_guard<std::mutex> lk(mu);
            rc = ZGPU_E_NODEVICE;
        }
    };
    std::vector<std::thread> th;
    for (size_t d = 0; d < j->devs.size(); ++d) {
        std::lock_guard<std::mutex> ctx_lock(j->devs[d].ctx->mu);  // the slots' streams are the context's
        size_t mine = 0;
        for (const SubBatch &sb : j->subs) mine += sb.dev_index == (int)d && sb.out_bytes;
        if (mine <= 1 && j->devs.size() == 1) {
            worker((int)d, 0, 1);
        } else {
            std::vector<std::thread> t2;
            t2.emplace_back(worker, (int)d, 1, 2);
            worker((int)d, 0, 2);
            t2[0].join();
        }
    }
    (void)hipSetDevice(prev_dev);
    return rc;
}

int zgpu_deflate_batch_begin(const uint8_t *in, const uint64_t *in_off, uint32_t n_units, int ctype, int lazy,
                             uint32_t flags, zgpu_job **job_out, uint64_t *out_total) {
    if (!in_off || !job_out || !out_total) return ZGPU_E_ARG;
    int rc = host_batch_compute(in, in_off, n_units, ctype, lazy, flags, nullptr, 0, job_out);
    if (rc) return rc;
    *out_total = (*job_out)->out_off[n_units];
    return ZGPU_OK;
}

int zgpu_deflate_batch_finish(zgpu_job *j, uint8_t *out, uint64_t out_cap, uint64_t *out_off, uint32_t *crc32,
                              uint32_t *adler32) {
    if (!j) return ZGPU_E_ARG;
    int rc = ZGPU_OK;
    const size_t n = j->n_units;
    const uint64_t total = j->out_off[n];
    if (!out_off || (total && !out) || (crc32 && j->crc.empty()) || (adler32 && j->adler.empty())) {
        rc = ZGPU_E_ARG;
    } else if (total > out_cap) {
        rc = ZGPU_E_OUTCAP;
    } else {
        rc = total ? host_batch_copy_out(j, out) : ZGPU_OK;
        if (rc == ZGPU_OK) {
            memcpy(out_off, j->out_off.data(), (n + 1) * sizeof(uint64_t));
            if (crc32) memcpy(crc32, j->crc.data(), n * sizeof(uint32_t));
            if (adler32) memcpy(adler32, j->adler.data(), n * sizeof(uint32_t));
        }
    }
    zgpu_job_release(j);
    return rc;
}

int zgpu_deflate_batch_finish_spaced(zgpu_job *j, uint8_t *out, uint64_t out_cap, const uint32_t *gap,
                                     uint64_t *out_pos, uint32_t *crc32, uint32_t *adler32) {
    if (!j) return ZGPU_E_ARG;
    int rc = ZGPU_OK;
    const size_t n = j->n_units;
    const uint64_t total = j->out_off[n];
    uint64_t gaps = 0;
    if (gap)
        for (size_t i = 0; i < n; ++i) gaps += gap[i];
    if (!gap || !out_pos || !out || (crc32 && j->crc.empty()) || (adler32 && j->adler.empty())) {
        rc = ZGPU_E_ARG;
    } else if (total + gaps > out_cap) {
        rc = ZGPU_E_OUTCAP;
    } else {
        uint64_t run = 0;
        for (size_t i = 0; i < n; ++i) {
            run += gap[i];
            out_pos[i] = j->out_off[i] + run;
        }
        out_pos[n] = total + run;
        bool all_staged = true;
        for (const zgpu_job::Dev &dv : j->devs) all_staged = all_staged && dv.staged;
        // where unit i's bytes lie now: in the page-locked staging of its device, or -- results too large
        // to stage -- in a contiguous host copy made first
        std::unique_ptr<uint8_t[]> flat;
        if (!all_staged && total) {
            flat.reset(new (std::nothrow) uint8_t[total]);
            rc = flat ? host_batch_copy_out(j, flat.get()) : ZGPU_E_NOMEM;
        }
        if (rc == ZGPU_OK && total) {
            const int nth = total >= (64ull << 20) ? 4 : (total >= (4ull << 20) ? 2 : 1);
            auto mover = [&](int part) {
                for (size_t si = (size_t)part; si < j->subs.size(); si += (size_t)nth) {
                    const SubBatch &sb = j->subs[si];
                    const uint8_t *src = flat ? flat.get() + j->out_off[sb.first]
                                              : (const uint8_t *)j->devs[sb.dev_index].h_res.p + sb.res_off;
                    const uint64_t base = j->out_off[sb.first];
                    for (uint32_t i = sb.first; i < sb.end; ++i)
                        memcpy(out + out_pos[i], src + (j->out_off[i] - base), j->out_off[i + 1] - j->out_off[i]);
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nth; ++t) th.emplace_back(mover, t);
            mover(0);
            for (auto &t : th) t.join();
        }
        if (rc == ZGPU_OK) {
            if (crc32) memcpy(crc32, j->crc.data(), n * sizeof(uint32_t));
            if (adler32) memcpy(adler32, j->adler.data(), n * sizeof(uint32_t));
        }
    }
    zgpu_job_release(j);
    return rc;
}

void zgpu_job_release(zgpu_job *j) {
    if (!j) return;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (zgpu_job::Dev &dv : j->devs) {
        // the result buffer goes back to its device as the spare one (or is released if the
        // device already has a spare at least as large)
        if (dv.d_res.p) {
            bool kept = false;
            if (dv.ctx) {
                std::lock_guard<std::mutex> lk(g_mu);
                if (dv.ctx->d_res.cap < dv.d_res.cap) {
                    std::swap(dv.ctx->d_res, dv.d_res);
                    kept = dv.d_res.p == nullptr;
                }
            }
            if (!kept && dv.d_res.p && hipSetDevice(dv.device) == hipSuccess) dv.d_res.release();
        }
        // the page-locked staging likewise (whether or not the device buffer came about: a failed
        // allocation there must not leak it), but only up to kSpareHostBytes: page-locked memory
        // cannot be swapped, and one large archive should not pin gigabytes for the life of the process
        if (dv.h_res.p) {
            bool kept_h = false;
            if (dv.ctx && dv.h_res.cap <= kSpareHostBytes) {
                std::lock_guard<std::mutex> lk(g_mu);
                if (dv.ctx->h_res.cap < dv.h_res.cap) {
                    std::swap(dv.ctx->h_res, dv.h_res);
                    kept_h = dv.h_res.p == nullptr;
                }
            }
            if (!kept_h) dv.h_res.release();
        }
    }
    if (have_cur) (void)hipSetDevice(cur);
    --g_live;
    delete j;
}

int zgpu_deflate_batch(const uint8_t *in, const uint64_t *in_off, uint32_t n_units, int ctype, int lazy,
                       uint32_t flags, uint8_t *out, uint64_t out_cap, uint64_t *out_off,
                       uint32_t *crc32, uint32_t *adler32) {
    if (!in_off || !out || !out_off) return ZGPU_E_ARG;
    if (crc32) flags |= ZGPU_F_WANT_CRC;
    if (adler32) flags |= ZGPU_F_WANT_ADLER;
    zgpu_job *job = nullptr;
    int rc = host_batch_compute(in, in_off, n_units, ctype, lazy, flags, out, out_cap, &job);
    if (rc) return rc;
    const uint64_t total = job->out_off[n_units];
    if (job->devs.size() > 1) {
        // several devices: nothing was copied yet (a device's place in out depends on the ones before it)
        if (total > out_cap) rc = ZGPU_E_OUTCAP;
        else if (total) rc = host_batch_copy_out(job, out);
    }
    if (rc == ZGPU_OK) {
        memcpy(out_off, job->out_off.data(), ((size_t)n_units + 1) * sizeof(uint64_t));
        if (crc32) memcpy(crc32, job->crc.data(), (size_t)n_units * sizeof(uint32_t));
        if (adler32) memcpy(adler32, job->adler.data(), (size_t)n_units * sizeof(uint32_t));
    }
    zgpu_job_release(job);
    return rc;
}

// ---- checksums -------------------------------------------------------------
uint32_t zgpu_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    return h_gf_mul(crc_a, h_xpow_bytes(len_b)) ^ crc_b;
}

uint32_t zgpu_adler32_combine(uint32_t a, uint32_t b, uint64_t len_b) {
    const uint32_t M = 65521u;
    const uint64_t rem = len_b % M;
    uint64_t s1a = a & 0xffff, s2a = a >> 16, s1b = b & 0xffff, s2b = b >> 16;
    // s1 = s1a + s1b - 1 ; s2 = s2a + s2b + len_b*(s1a - 1)
    uint64_t s1 = (s1a + s1b + M - 1) % M;
    uint64_t s2 = (s2a + s2b + rem * ((s1a + M - 1) % M)) % M;
    return (uint32_t)((s2 << 16) | s1);
}

int zgpu_checksum_device(const void *d_in, uint64_t len, uint32_t *crc32, uint32_t *adler32,
                         void *stream_) {
    if (!d_in && len) return ZGPU_E_ARG;
    DeviceState *ds = nullptr;
    int rc = get_state(&ds);
    if (rc) return rc;
    std::lock_guard<std::mutex> ck_lock(ds->ck_mu);
    if (!ds->ck) {
        ds->ck = new (std::nothrow) CkScratch();
        if (!ds->ck) return ZGPU_E_NOMEM;
    }
    CkScratch &K = *ds->ck;
    hipStream_t st = (hipStream_t)stream_;
    // large buffers go through the wide kernel (one workgroup per CU, 4 MiB segments)
    const bool wide = len >= ZGPU_CKW_MIN_BYTES;
    const uint64_t seg = wide ? ZGPU_CKW_SEG : ZGPU_CK_SEG;
    const uint64_t nseg = len ? (len + seg - 1) / seg : 1;
    if ((rc = K.seg_off.ensure((nseg + 1) * 8)) || (rc = K.seg_crc.ensure(nseg * 4)) ||
        (rc = K.seg_adler.ensure(nseg * 4)) || (rc = K.out2.ensure(8)))
        return rc;
    // segment offsets are written and the per-segment values combined on the device: the host
    // only waits for two words
    hipLaunchKernelGGL(zgpu_na_offsets_kernel, dim3((uint32_t)(nseg / 256 + 1)), dim3(256), 0, st,
                       K.seg_off.as<uint64_t>(), (uint32_t)nseg, seg, len);
    HIPCHK(hipEventRecord(ds->ev0, st));
    if (wide) {
        const int grid = (int)(nseg < (uint64_t)ds->cus ? nseg : (uint64_t)ds->cus);
        hipLaunchKernelGGL(zgpu_checksum_wide_kernel, dim3(grid), dim3(ZGPU_CKW_THREADS), 0, st,
                           (const uint8_t *)d_in, K.seg_off.as<uint64_t>(), (uint32_t)nseg, ds->d_tables_wide,
                           K.seg_crc.as<uint32_t>(), K.seg_adler.as<uint32_t>());
    } else {
        const int grid = (int)(nseg < (uint64_t)ds->cus * 8 ? nseg : (uint64_t)ds->cus * 8);
        hipLaunchKernelGGL(zgpu_checksum_kernel, dim3(grid), dim3(ZGPU_CK_THREADS), 0, st,
                           (const uint8_t *)d_in, K.seg_off.as<uint64_t>(), (uint32_t)nseg, ds->d_tables,
                           K.seg_crc.as<uint32_t>(), K.seg_adler.as<uint32_t>());
    }
    HIPCHK(hipEventRecord(ds->ev1, st));
    hipLaunchKernelGGL(zgpu_na_fold_kernel, dim3(1), dim3(256), 0, st, K.seg_off.as<uint64_t>(),
                       K.seg_crc.as<uint32_t>(), K.seg_adler.as<uint32_t>(), (uint32_t)nseg, K.out2.as<uint32_t>());
    HIPCHK(hipGetLastError());
    uint32_t r[2] = {0, 1};
    hipError_t e = hipMemcpyAsync(r, K.out2.p, 8, hipMemcpyDeviceToHost, st);
    const hipError_t e2 = hipStreamSynchronize(st);  // also when the copy failed to enqueue: nothing may still write r
    if (e == hipSuccess) e = e2;
    HIPCHK(e);
    HIPCHK(hipEventElapsedTime(&ds->last_ck_ms, ds->ev0, ds->ev1));
    if (crc32) *crc32 = r[0];
    if (adler32) *adler32 = r[1];
    return ZGPU_OK;
}

int zgpu_noalloc_setup(void) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return ZGPU_E_ARG;
    std::lock_guard<std::mutex> lk(g_na_mu);
    if (g_na_ptr[dev]) return ZGPU_OK;
    unsigned char *ptr = nullptr;
    HIPCHK(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    std::vector<uint32_t> t;
    build_tables(t);
    HIPCHK(hipMemcpy(ptr + kNaTables, t.data(), kCkTabWords * 4, hipMemcpyHostToDevice));  // the workgroup kernel's part
    g_na_ptr[dev] = ptr;
    return ZGPU_OK;
}

uint64_t zgpu_noalloc_max_bytes(void) { return (uint64_t)kNaMaxSegs * ZGPU_CK_SEG; }

int zgpu_checksum_device_noalloc(const void *d_in, uint64_t len, uint32_t *d_out2, void *stream_) {
    if (!d_out2 || (!d_in && len)) return ZGPU_E_ARG;
    if (len > zgpu_noalloc_max_bytes()) return ZGPU_E_TOOBIG;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return ZGPU_E_ARG;
    unsigned char *ptr;
    {
        std::lock_guard<std::mutex> lk(g_na_mu);
        ptr = g_na_ptr[dev];
    }
    if (!ptr) {
        // first call on this device: one blocking table upload (zgpu_noalloc_setup moves it
        // out of, say, a stream capture)
        int rc = zgpu_noalloc_setup();
        if (rc) return rc;
        std::lock_guard<std::mutex> lk(g_na_mu);
        ptr = g_na_ptr[dev];
    }
    hipStream_t st = (hipStream_t)stream_;
    const uint64_t seg = ZGPU_CK_SEG;
    const uint32_t nseg = len ? (uint32_t)((len + seg - 1) / seg) : 1u;
    uint64_t *seg_off = (uint64_t *)(ptr + kNaSegOff);
    uint32_t *seg_crc = (uint32_t *)(ptr + kNaCrc), *seg_adler = (uint32_t *)(ptr + kNaAdler);
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipLaunchKernelGGL(zgpu_na_offsets_kernel, dim3(nseg / 256 + 1), dim3(256), 0, st, seg_off, nseg, seg, len);
    const uint32_t grid = nseg < (uint32_t)cus * 8u ? nseg : (uint32_t)cus * 8u;
    hipLaunchKernelGGL(zgpu_checksum_kernel, dim3(grid), dim3(ZGPU_CK_THREADS), 0, st, (const uint8_t *)d_in,
                       seg_off, nseg, (const uint32_t *)(ptr + kNaTables), seg_crc, seg_adler);
    hipLaunchKernelGGL(zgpu_na_fold_kernel, dim3(1), dim3(256), 0, st, seg_off, seg_crc, seg_adler, nseg, d_out2);
    HIPCHK(hipGetLastError());
    return ZGPU_OK;
}

int zgpu_checksum_kernel_ms(float *ms) {
    DeviceState *ds = nullptr;
    int rc = get_state(&ds);
    if (rc) return rc;
    if (ms) *ms = ds->last_ck_ms;
    return ZGPU_OK;
}

int zgpu_checksum(const uint8_t *in, uint64_t len, uint32_t *crc32, uint32_t *adler32) {
    if (!in && len) return ZGPU_E_ARG;
    HostDeviceScope scope;
    if (scope.rc) return scope.rc;
    DeviceState *ds = nullptr;
    int rc = get_state(&ds);
    if (rc) return rc;
    HostCtx *ctx = nullptr;
    if ((rc = get_host_ctx(ds, &ctx)) != ZGPU_OK) return rc;
    std::lock_guard<std::mutex> ctx_lock(ctx->mu);
    // chunks of kSubBytes through two of the device's slots: the upload of one chunk runs while the
    // other is summed; per-chunk values are folded on the device and combined here at the end
    const uint64_t nchunks = len ? (len + kSubBytes - 1) / kSubBytes : 1;
    if ((rc = ctx->ck_res.ensure(nchunks * 8)) != ZGPU_OK) return rc;
    for (uint64_t c = 0; c < nchunks; ++c) {
        Slot &sl = ctx->slot[c & 1];
        const uint64_t off = c * kSubBytes, bytes = len - off < kSubBytes ? len - off : kSubBytes;
        const uint64_t lead = (uintptr_t)(in + off) & 255u;
        const uint64_t nseg = bytes ? (bytes + ZGPU_CK_SEG - 1) / ZGPU_CK_SEG : 1;
        if ((rc = sl.d_in.ensure(lead + bytes + 64)) || (rc = sl.ck_off.ensure((nseg + 1) * 8)) ||
            (rc = sl.ck_crc.ensure(nseg * 4)) || (rc = sl.ck_adler.ensure(nseg * 4)))
            break;
        uint8_t *d = sl.d_in.as<uint8_t>() + lead;
        if (bytes) HIPCHK(hipMemcpyAsync(d, in + off, bytes, hipMemcpyHostToDevice, sl.st));
        hipLaunchKernelGGL(zgpu_na_offsets_kernel, dim3((uint32_t)(nseg / 256 + 1)), dim3(256), 0, sl.st,
                           sl.ck_off.as<uint64_t>(), (uint32_t)nseg, (uint64_t)ZGPU_CK_SEG, bytes);
        const uint32_t grid = (uint32_t)(nseg < (uint64_t)ds->cus * 8 ? nseg : (uint64_t)ds->cus * 8);
        hipLaunchKernelGGL(zgpu_checksum_kernel, dim3(grid), dim3(ZGPU_CK_THREADS), 0, sl.st, d,
                           sl.ck_off.as<uint64_t>(), (uint32_t)nseg, ds->d_tables, sl.ck_crc.as<uint32_t>(),
                           sl.ck_adler.as<uint32_t>());
        hipLaunchKernelGGL(zgpu_na_fold_kernel, dim3(1), dim3(256), 0, sl.st, sl.ck_off.as<uint64_t>(),
                           sl.ck_crc.as<uint32_t>(), sl.ck_adler.as<uint32_t>(), (uint32_t)nseg,
                           ctx->ck_res.as<uint32_t>() + 2 * c);
    }
    const hipError_t e0 = hipStreamSynchronize(ctx->slot[0].st), e1 = hipStreamSynchronize(ctx->slot[1].st);
    if (rc) return rc;
    HIPCHK(e0);
    HIPCHK(e1);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> r;
    try {
        r.resize(nchunks * 2);
    } catch (const std::bad_alloc &) {
        return ZGPU_E_NOMEM;
    }
    HIPCHK(hipMemcpyAsync(r.data(), ctx->ck_res.p, nchunks * 8, hipMemcpyDeviceToHost, ctx->slot[0].st));
    HIPCHK(hipStreamSynchronize(ctx->slot[0].st));
    uint32_t crc = r[0], ad = r[1];
    for (uint64_t c = 1; c < nchunks; ++c) {
        const uint64_t off = c * kSubBytes, bytes = len - off < kSubBytes ? len - off : kSubBytes;
        crc = zgpu_crc32_combine(crc, r[2 * c], bytes);
        ad = zgpu_adler32_combine(ad, r[2 * c + 1], bytes);
    }
    if (crc32) *crc32 = crc;
    if (adler32) *adler32 = ad;
    return ZGPU_OK;
}

int zgpu_crc32_update(const uint8_t *in, uint64_t len, uint32_t crc, uint32_t *out) {
    uint32_t c = 0;
    int rc = zgpu_checksum(in, len, &c, nullptr);
    if (rc) return rc;
    if (out) *out = zgpu_crc32_combine(crc, c, len);
    return ZGPU_OK;
}

int zgpu_adler32_update(const uint8_t *in, uint64_t len, uint32_t adler, uint32_t *out) {
    uint32_t a = 1;
    int rc = zgpu_checksum(in, len, nullptr, &a);
    if (rc) return rc;
    if (out) *out = zgpu_adler32_combine(adler, a, len);
    return ZGPU_OK;
}

}  // extern "C"
