// Host side of libzgpu.so: the C ABI declared in include/zgpu.h.
// Thin by design -- unit tables, workspace, launches and copies.  No CPU
// compression or checksum path exists here: without a GPU every entry point
// returns ZGPU_E_NODEVICE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/zgpu.h"
#include "zgpu_dev.h"
#include "deflate_kernel.hip"
#include "checksum_kernel.hip"

namespace {

thread_local int g_last_hip = 0;
int g_max_chain = 8;
int g_nice = 32;
uint64_t g_split_bytes = 1ull << 20;  // units longer than twice this are cut into pieces of at most this size
int g_deep = 0;        // long-match steps of the search kernel: deep-walk budget ...
int g_fan = 1;         // ... and probe-ahead fans per item
int g_min_lazy = 258;  // the parse always looks one position ahead unless told otherwise
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

struct DeviceState {
    int device = -1;
    int cus = 0;
    uint32_t *d_tables = nullptr;  // checksum kernel tables
    uint32_t *d_tables_wide = nullptr;  // 11-bit tables of the wide checksum kernel
    // scratch for zgpu_checksum_device
    uint64_t *d_seg_off = nullptr;
    uint32_t *d_seg_crc = nullptr, *d_seg_adler = nullptr;
    uint64_t seg_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ck_ms = 0.f;
    std::mutex ck_mu;  // zgpu_checksum_device calls on one device share the scratch above
};
std::mutex g_mu;
std::vector<DeviceState *> g_devs;
std::vector<int> g_allowed;  // devices named by zgpu_init

void build_tables(std::vector<uint32_t> &t) {
    t.assign(20 * 256 + 8, 0);
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
}

// tables of zgpu_checksum_wide_kernel, derived from the byte-sliced ones by linearity
void build_tables_wide(const std::vector<uint32_t> &t, std::vector<uint32_t> &w) {
    w.assign(15 * 2048 + 10 + 256, 0);
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
    const uint32_t adv = h_xpow_bytes(16ull * ZGPU_CKW_THREADS);
    for (int c = 0; c < 3; ++c)
        for (uint32_t v = 0; v < 2048; ++v) {
            const uint64_t sh = (uint64_t)v << (11 * c);
            w[(12 + c) * 2048 + v] = sh >> 32 ? 0u : h_gf_mul((uint32_t)sh, adv);
        }
    for (int lv = 0; lv < 10; ++lv) w[15 * 2048 + lv] = h_xpow_bytes(16ull << lv);
    for (int i = 0; i < 256; ++i) w[15 * 2048 + 10 + i] = t[i];
}

void free_state(DeviceState *s) {
    (void)hipFree(s->d_tables);
    (void)hipFree(s->d_tables_wide);
    (void)hipFree(s->d_seg_off);
    (void)hipFree(s->d_seg_crc);
    (void)hipFree(s->d_seg_adler);
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

}  // namespace

struct zgpu_plan {
    DeviceState *ds = nullptr;
    uint32_t n_units = 0;
    int ctype = 2, lazy = 0;
    uint32_t flags = 0;
    uint64_t in_total = 0, out_bound = 0, tmp_bytes = 0;
    uint32_t n_pieces = 0;           // what the kernels see: units, the long ones cut into pieces
    bool split = false;              // some unit has more than one piece
    uint32_t *d_prime = nullptr;     // per piece: bytes before it that prime its window
    uint8_t *d_final = nullptr;      // per piece: ends its stream
    uint64_t *d_unit_in_off = nullptr, *d_unit_out_off = nullptr;  // per unit (only when split)
    uint32_t *d_first_piece = nullptr;
    std::vector<uint32_t> first_piece;
    uint64_t *d_in_off = nullptr, *d_slot_off = nullptr, *d_out_off = nullptr;
    uint32_t *d_out_size = nullptr, *d_status = nullptr, *d_crc = nullptr, *d_adler = nullptr;
    uint16_t *d_tok = nullptr;       // token slots of one round
    uint32_t *d_counter = nullptr;
    uint64_t *d_tok_off = nullptr;   // per unit: first token within its round's buffer
    uint32_t *d_blk_first = nullptr; // per unit: index of its first block
    uint32_t *d_hist = nullptr, *d_ntok = nullptr;  // per block
    uint8_t *d_out_tmp = nullptr;
    struct Round { uint32_t first, end; };
    std::vector<Round> rounds;       // units whose tokens share the token buffer at one time
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ran = false;
};

// a host-buffer batch between its two calls: results fetched, compressed bytes still on the device
struct zgpu_job {
    int device = -1;
    zgpu_plan *plan = nullptr;
    uint8_t *d_in = nullptr, *d_out = nullptr;
    std::vector<uint64_t> out_off;
    std::vector<uint32_t> crc, adler;
};

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
        // x^(8 * after) by square and multiply
        uint32_t r = 0x80000000u, sq = 0x00800000u;
        for (uint64_t e = after; e; e >>= 1) {
            if (e & 1) r = gf_mul(r, sq);
            sq = gf_mul(sq, sq);
        }
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

namespace {
uint64_t unit_blocks(uint64_t n) { return n ? (n + ZGPU_BLOCK_BYTES - 1) / ZGPU_BLOCK_BYTES : 1; }
// token slots (16 bits each) reserved for a unit: full blocks at their fixed stride, the last
// one exact, rounded so that every region starts on 8 bytes and reads of four slots stay inside
uint64_t unit_tok_cap(uint64_t n) {
    const uint64_t nb = unit_blocks(n), last = n - (nb - 1) * ZGPU_BLOCK_BYTES;
    return (nb - 1) * ZGPU_BLOCK_TOK + ((last + 1 + 3) & ~3ull) + 4;
}
}  // namespace

extern "C" {

int zgpu_init(const int *device_ids, int n_devices) {
    int count = 0;
    HIPCHK(hipGetDeviceCount(&count));
    if (count <= 0) return ZGPU_E_NODEVICE;
    std::lock_guard<std::mutex> lk(g_mu);
    g_allowed.clear();
    for (int i = 0; i < n_devices; ++i) {
        if (device_ids[i] < 0 || device_ids[i] >= count) return ZGPU_E_ARG;
        g_allowed.push_back(device_ids[i]);
    }
    return ZGPU_OK;
}

void zgpu_shutdown(void) {
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
int zgpu_prof_read(unsigned long long *out8, int reset) {
    if (!g_prof) {
        HIPCHK(hipMalloc(&g_prof, 128));
        HIPCHK(hipMemset(g_prof, 0, 128));
    }
    HIPCHK(hipDeviceSynchronize());
    if (out8) HIPCHK(hipMemcpy(out8, g_prof, 128, hipMemcpyDeviceToHost));
    if (reset) HIPCHK(hipMemset(g_prof, 0, 128));
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
    g_deep = deep_chain;
    g_fan = probe_fans;
    return ZGPU_OK;
}

int zgpu_set_split(uint64_t piece_bytes) {
    if (piece_bytes && piece_bytes < (1ull << 20)) return ZGPU_E_ARG;
    g_split_bytes = piece_bytes;
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
    for (uint32_t i = 0; i < n_units; ++i) {
        if (in_off[i + 1] < in_off[i]) return ZGPU_E_ARG;
        if (in_off[i + 1] - in_off[i] > ZGPU_MAX_UNIT_BYTES) return ZGPU_E_TOOBIG;
    }
    DeviceState *ds = nullptr;
    int rc = get_state(&ds);
    if (rc) return rc;
    auto *p = new (std::nothrow) zgpu_plan();
    if (!p) return ZGPU_E_NOMEM;
    p->ds = ds;
    p->n_units = n_units;
    p->ctype = ctype;
    p->lazy = lazy;
    p->flags = flags;
    p->in_total = in_off[n_units];
    // ---- pieces: what the kernels work on.  A unit longer than twice the split size is cut
    // into pieces of that size, each compressed by its own workgroup as a chunk of the unit's
    // stream: byte-aligned end (empty stored block), window primed with the 32 KiB before it.
    // A unit's pieces are consecutive, so its output is their concatenation.
    const bool chunked = (flags & ZGPU_F_SYNC_ALIGN) && !(flags & ZGPU_F_FINAL_EACH);
    const bool prime_units = chunked && (flags & ZGPU_F_PRIME_WINDOW);
    // Piece size: a batch with fewer units than kPiecesPerCu per CU (the unit queue then balances
    // poorly at its tail, and the emit kernel's workgroups run short) cuts units longer than two
    // pieces into pieces of total / (kPiecesPerCu x CUs) bytes, between 128 KiB and the set limit.
    constexpr uint32_t kPiecesPerCu = 16;
    uint64_t piece = 0;
    if (ctype != ZGPU_NONE && g_split_bytes && n_units < kPiecesPerCu * (uint32_t)ds->cus) {
        piece = align_up((in_off[n_units] - in_off[0]) / ((uint64_t)kPiecesPerCu * (uint64_t)ds->cus) + 1, 65536);
        if (piece < (128u << 10)) piece = 128u << 10;
        if (piece > g_split_bytes) piece = g_split_bytes;
    }
    std::vector<uint64_t> pc_off;        // piece offsets into the input, n_pieces + 1
    std::vector<uint32_t> first_piece;   // per unit, n_units + 1
    std::vector<uint32_t> pc_prime;      // bytes of the same stream before the piece that prime its window
    std::vector<uint8_t> pc_final;       // piece ends its stream (BFINAL) / ends with an empty stored block
    for (uint32_t i = 0; i < n_units; ++i) {
        const uint64_t o = in_off[i], n = in_off[i + 1] - o;
        const uint64_t split = (piece && n > 2 * piece) ? piece : 0;
        const uint64_t np = split ? (n + split - 1) / split : 1;
        const bool unit_final = !chunked || ((flags & ZGPU_F_FINAL_LAST) && i == n_units - 1);
        first_piece.push_back((uint32_t)pc_off.size());
        for (uint64_t k = 0; k < np; ++k) {
            pc_off.push_back(o + k * split);
            uint64_t before = k ? k * split : (prime_units ? o - in_off[0] : 0);
            if (before > 32768) before = 32768;
            pc_prime.push_back((uint32_t)before & ~63u);
            pc_final.push_back(unit_final && k == np - 1);
        }
        if (pc_off.size() > 0x7fffffffull) {
            delete p;
            return ZGPU_E_TOOBIG;
        }
    }
    const uint32_t n_pieces = (uint32_t)pc_off.size();
    first_piece.push_back(n_pieces);
    pc_off.push_back(in_off[n_units]);
    p->n_pieces = n_pieces;
    p->split = n_pieces != n_units;
    std::vector<uint64_t> slot(n_pieces + 1);
    uint64_t acc = 0;
    for (uint32_t i = 0; i < n_pieces; ++i) {
        slot[i] = acc;
        acc += align_up(zgpu_deflate_bound(pc_off[i + 1] - pc_off[i]) + 32, 16);
    }
    slot[n_pieces] = acc;
    p->tmp_bytes = acc;
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
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) cap_bytes = free_b / 3;
        if (const char *e = getenv("ZGPU_TOK_BYTES")) cap_bytes = strtoull(e, nullptr, 10);
        const uint64_t cap_tokens = cap_bytes / 2;
        uint64_t used = 0;
        uint32_t first = 0;
        for (uint32_t i = 0; i < n_pieces; ++i) {
            const uint64_t n = pc_off[i + 1] - pc_off[i], need = unit_tok_cap(n);
            if (used && used + need > cap_tokens) {
                p->rounds.push_back({first, i});
                first = i;
                used = 0;
            }
            tok_off[i] = used;
            used += need;
            if (used > tok_buf_tokens) tok_buf_tokens = used;
            if (n_blocks + unit_blocks(n) > 0xffffffffull) {
                delete p;
                return ZGPU_E_TOOBIG;
            }
            blk_first[i] = (uint32_t)n_blocks;
            n_blocks += unit_blocks(n);
        }
        p->rounds.push_back({first, n_pieces});
    }
#define PCHK(expr)                 \
    do {                           \
        hipError_t e_ = (expr);    \
        if (e_ != hipSuccess) {    \
            g_last_hip = (int)e_;  \
            zgpu_plan_destroy(p);  \
            return e_ == hipErrorOutOfMemory ? ZGPU_E_NOMEM : ZGPU_E_NODEVICE; \
        }                          \
    } while (0)
    const size_t offb = (size_t)(n_pieces + 1) * 8, uoffb = (size_t)(n_units + 1) * 8;
    PCHK(hipMalloc(&p->d_in_off, offb));
    PCHK(hipMalloc(&p->d_slot_off, offb));
    PCHK(hipMalloc(&p->d_out_off, offb));
    PCHK(hipMalloc(&p->d_out_size, (size_t)n_pieces * 4));
    PCHK(hipMalloc(&p->d_status, (size_t)n_pieces * 4));
    PCHK(hipMalloc(&p->d_prime, (size_t)n_pieces * 4));
    PCHK(hipMalloc(&p->d_final, (size_t)n_pieces));
    PCHK(hipMalloc(&p->d_crc, (size_t)n_units * 4));
    PCHK(hipMalloc(&p->d_adler, (size_t)n_units * 4));
    PCHK(hipMalloc(&p->d_counter, (p->rounds.size() * 2 + 2) * 4));
    PCHK(hipMalloc(&p->d_out_tmp, p->tmp_bytes + 64));
    if (ctype != ZGPU_NONE) {
        PCHK(hipMalloc(&p->d_tok, (size_t)tok_buf_tokens * 2 + 64));
        PCHK(hipMalloc(&p->d_tok_off, (size_t)n_pieces * 8));
        PCHK(hipMalloc(&p->d_blk_first, (size_t)n_pieces * 4));
        PCHK(hipMalloc(&p->d_hist, (size_t)n_blocks * 320 * 4));
        PCHK(hipMalloc(&p->d_ntok, (size_t)n_blocks * 4));
        PCHK(hipMemcpy(p->d_tok_off, tok_off.data(), (size_t)n_pieces * 8, hipMemcpyHostToDevice));
        PCHK(hipMemcpy(p->d_blk_first, blk_first.data(), (size_t)n_pieces * 4, hipMemcpyHostToDevice));
    }
    PCHK(hipMemcpy(p->d_in_off, pc_off.data(), offb, hipMemcpyHostToDevice));
    PCHK(hipMemcpy(p->d_slot_off, slot.data(), offb, hipMemcpyHostToDevice));
    PCHK(hipMemcpy(p->d_prime, pc_prime.data(), (size_t)n_pieces * 4, hipMemcpyHostToDevice));
    PCHK(hipMemcpy(p->d_final, pc_final.data(), (size_t)n_pieces, hipMemcpyHostToDevice));
    if (p->split) {
        // the caller's view: unit offsets into the input, first piece of every unit, output offsets
        PCHK(hipMalloc(&p->d_unit_in_off, uoffb));
        PCHK(hipMalloc(&p->d_first_piece, (size_t)(n_units + 1) * 4));
        PCHK(hipMalloc(&p->d_unit_out_off, uoffb));
        PCHK(hipMemcpy(p->d_unit_in_off, in_off, uoffb, hipMemcpyHostToDevice));
        PCHK(hipMemcpy(p->d_first_piece, first_piece.data(), (size_t)(n_units + 1) * 4, hipMemcpyHostToDevice));
        p->first_piece = first_piece;
    }
    PCHK(hipMemset(p->d_crc, 0, (size_t)n_units * 4));
    PCHK(hipMemset(p->d_adler, 0, (size_t)n_units * 4));
    for (int i = 0; i < 5; ++i) PCHK(hipEventCreate(&p->ev[i]));
#undef PCHK
    *out = p;
    return ZGPU_OK;
}

void zgpu_plan_destroy(zgpu_plan *p) {
    if (!p) return;
    (void)hipFree(p->d_in_off);
    (void)hipFree(p->d_slot_off);
    (void)hipFree(p->d_out_off);
    (void)hipFree(p->d_out_size);
    (void)hipFree(p->d_status);
    (void)hipFree(p->d_crc);
    (void)hipFree(p->d_adler);
    (void)hipFree(p->d_counter);
    (void)hipFree(p->d_out_tmp);
    (void)hipFree(p->d_tok);
    (void)hipFree(p->d_tok_off);
    (void)hipFree(p->d_blk_first);
    (void)hipFree(p->d_hist);
    (void)hipFree(p->d_ntok);
    (void)hipFree(p->d_prime);
    (void)hipFree(p->d_final);
    (void)hipFree(p->d_unit_in_off);
    (void)hipFree(p->d_unit_out_off);
    (void)hipFree(p->d_first_piece);
    for (int i = 0; i < 5; ++i)
        if (p->ev[i]) (void)hipEventDestroy(p->ev[i]);
    delete p;
}

uint64_t zgpu_plan_out_bound(const zgpu_plan *p) { return p ? p->out_bound : 0; }

int zgpu_plan_run(zgpu_plan *p, const void *d_in, void *d_out, uint64_t out_cap, void *stream_) {
    if (!p || !d_out || (!d_in && p->in_total)) return ZGPU_E_ARG;
    hipStream_t st = (hipStream_t)stream_;
    HIPCHK(hipEventRecord(p->ev[0], st));
    HIPCHK(hipMemsetAsync(p->d_counter, 0, (p->rounds.size() * 2 + 2) * 4, st));
    HIPCHK(hipEventRecord(p->ev[1], st));
    if (p->ctype == ZGPU_NONE) {
        int grid = p->n_units < (uint32_t)p->ds->cus * 8 ? (int)p->n_units : p->ds->cus * 8;
        hipLaunchKernelGGL(zgpu_stored_kernel, dim3(grid), dim3(256), 0, st, (const uint8_t *)d_in,
                           p->d_in_off, p->d_slot_off, p->d_out_tmp, p->d_out_size, p->d_status,
                           p->n_units, p->flags);
        HIPCHK(hipEventRecord(p->ev[4], st));
    } else {
        for (size_t r = 0; r < p->rounds.size(); ++r) {
            const uint32_t first = p->rounds[r].first, end = p->rounds[r].end, cnt = end - first;
            Params P;
            P.in = (const uint8_t *)d_in;
            P.in_off = p->d_in_off;
            P.tok = p->d_tok;
            P.tok_off = p->d_tok_off;
            P.blk_first = p->d_blk_first;
            P.hist = p->d_hist;
            P.ntok = p->d_ntok;
            P.counter = p->d_counter + 2 * r;
            P.unit_first = first;
            P.unit_end = end;
            P.lazy = p->lazy > g_min_lazy ? p->lazy : g_min_lazy;
            P.max_chain = g_max_chain;
            P.nice = g_nice;
            P.deep = g_deep;
            P.fan = g_fan;
            P.unit_prime = p->d_prime;
            P.prof = g_prof;
            const int sgrid = (int)(cnt < (uint32_t)p->ds->cus ? cnt : (uint32_t)p->ds->cus);
            if (g_nice > 32)
                hipLaunchKernelGGL(zgpu_deflate_thorough_kernel, dim3(sgrid), dim3(1024), 0, st, P);
            else
                hipLaunchKernelGGL(zgpu_deflate_kernel, dim3(sgrid), dim3(1024), 0, st, P);
            if (r == 0) HIPCHK(hipEventRecord(p->ev[4], st));
            EParams E;
            E.in_off = p->d_in_off;
            E.slot_off = p->d_slot_off;
            E.out_tmp = p->d_out_tmp;
            E.out_size = p->d_out_size;
            E.status = p->d_status;
            E.tok = p->d_tok;
            E.tok_off = p->d_tok_off;
            E.blk_first = p->d_blk_first;
            E.hist = p->d_hist;
            E.ntok = p->d_ntok;
            E.counter = p->d_counter + 2 * r + 1;
            E.unit_first = first;
            E.unit_end = end;
            E.unit_final = p->d_final;
            E.ctype = p->ctype;
            const uint32_t emax = (uint32_t)p->ds->cus * ZGPU_EMIT_WG_PER_CU;
            hipLaunchKernelGGL(zgpu_emit_kernel, dim3(cnt < emax ? cnt : emax), dim3(256), 0, st, E);
        }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(p->ev[2], st));
    hipLaunchKernelGGL(zgpu_scan_kernel, dim3(1), dim3(1024), 0, st, p->d_out_size, p->d_out_off,
                       p->n_pieces);
    {
        int grid = p->n_pieces < (uint32_t)p->ds->cus * 8 ? (int)p->n_pieces : p->ds->cus * 8;
        hipLaunchKernelGGL(zgpu_compact_kernel, dim3(grid), dim3(256), 0, st, p->d_out_tmp,
                           p->d_slot_off, p->d_out_off, (uint8_t *)d_out, out_cap, p->d_status,
                           p->n_pieces);
    }
    if (p->split)
        hipLaunchKernelGGL(zgpu_unit_offsets_kernel, dim3(p->n_units / 256 + 1), dim3(256), 0, st, p->d_out_off,
                           p->d_first_piece, p->d_unit_out_off, p->n_units);
    if (p->flags & (ZGPU_F_WANT_CRC | ZGPU_F_WANT_ADLER)) {
        int grid = p->n_units < (uint32_t)p->ds->cus * 8 ? (int)p->n_units : p->ds->cus * 8;
        hipLaunchKernelGGL(zgpu_checksum_kernel, dim3(grid), dim3(ZGPU_CK_THREADS), 0, st,
                           (const uint8_t *)d_in, p->split ? p->d_unit_in_off : p->d_in_off, p->n_units,
                           p->ds->d_tables, p->d_crc, p->d_adler);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(p->ev[3], st));
    p->ran = true;
    return ZGPU_OK;
}

const uint64_t *zgpu_plan_out_off_device(const zgpu_plan *p) {
    return p ? (p->split ? p->d_unit_out_off : p->d_out_off) : nullptr;
}
const uint32_t *zgpu_plan_crc_device(const zgpu_plan *p) { return p ? p->d_crc : nullptr; }
const uint32_t *zgpu_plan_adler_device(const zgpu_plan *p) { return p ? p->d_adler : nullptr; }

int zgpu_plan_fetch(zgpu_plan *p, uint64_t *out_off, uint32_t *crc32, uint32_t *adler32, void *stream_) {
    if (!p || !p->ran) return ZGPU_E_ARG;
    hipStream_t st = (hipStream_t)stream_;
    HIPCHK(hipStreamSynchronize(st));
    std::vector<uint32_t> status(p->n_pieces);
    HIPCHK(hipMemcpy(status.data(), p->d_status, (size_t)p->n_pieces * 4, hipMemcpyDeviceToHost));
    if (out_off)
        HIPCHK(hipMemcpy(out_off, p->split ? p->d_unit_out_off : p->d_out_off, (size_t)(p->n_units + 1) * 8,
                         hipMemcpyDeviceToHost));
    if (crc32) HIPCHK(hipMemcpy(crc32, p->d_crc, (size_t)p->n_units * 4, hipMemcpyDeviceToHost));
    if (adler32) HIPCHK(hipMemcpy(adler32, p->d_adler, (size_t)p->n_units * 4, hipMemcpyDeviceToHost));
    int rc = ZGPU_OK;
    for (uint32_t i = 0; i < p->n_pieces; ++i) {
        if (status[i] == 2) rc = ZGPU_E_OUTCAP;
        else if (status[i] && rc == ZGPU_OK) rc = ZGPU_E_UNIT;
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

int zgpu_deflate_batch_begin(const uint8_t *in, const uint64_t *in_off, uint32_t n_units, int ctype, int lazy,
                             uint32_t flags, zgpu_job **job_out, uint64_t *out_total) {
    if (!in_off || !job_out || !out_total) return ZGPU_E_ARG;
    *job_out = nullptr;
    HostDeviceScope scope;
    if (scope.rc) return scope.rc;
    auto *j = new (std::nothrow) zgpu_job();
    if (!j) return ZGPU_E_NOMEM;
    auto fail = [&](int code) {
        zgpu_job_release(j);
        return code;
    };
    if (hipGetDevice(&j->device) != hipSuccess) return fail(ZGPU_E_NODEVICE);
    int rc = zgpu_plan_create(&j->plan, in_off, n_units, ctype, lazy, flags);
    if (rc) return fail(rc);
    zgpu_plan *p = j->plan;
    const uint64_t base = in_off[0], total = in_off[n_units] - base;
    if (total && !in) return fail(ZGPU_E_ARG);
    hipError_t e;
    // The plan indexes its input with the caller's absolute offsets: only the bytes of the
    // units are allocated and copied, and the base pointer handed to the plan is moved back by
    // the first offset (kept congruent modulo 256 so the kernels see the caller's alignment).
    const uint64_t lead = base & 255u;
    if ((e = hipMalloc(&j->d_in, lead + total + 64)) != hipSuccess) { g_last_hip = e; return fail(ZGPU_E_NOMEM); }
    if ((e = hipMalloc(&j->d_out, p->out_bound + 64)) != hipSuccess) { g_last_hip = e; return fail(ZGPU_E_NOMEM); }
    if (total && (e = hipMemcpy(j->d_in + lead, in + base, total, hipMemcpyHostToDevice)) != hipSuccess) {
        g_last_hip = e;
        return fail(ZGPU_E_NODEVICE);
    }
    const uint8_t *d_in0 = (const uint8_t *)((uintptr_t)j->d_in + lead - base);
    rc = zgpu_plan_run(p, d_in0, j->d_out, p->out_bound, nullptr);
    if (rc) return fail(rc);
    try {
        j->out_off.resize((size_t)n_units + 1);
        if (flags & ZGPU_F_WANT_CRC) j->crc.resize(n_units);
        if (flags & ZGPU_F_WANT_ADLER) j->adler.resize(n_units);
    } catch (const std::bad_alloc &) {
        return fail(ZGPU_E_NOMEM);
    }
    rc = zgpu_plan_fetch(p, j->out_off.data(), j->crc.empty() ? nullptr : j->crc.data(),
                         j->adler.empty() ? nullptr : j->adler.data(), nullptr);
    if (rc) return fail(rc);
    // the workspace is no longer needed; the compressed bytes wait on the device
    zgpu_plan_destroy(j->plan);
    j->plan = nullptr;
    (void)hipFree(j->d_in);
    j->d_in = nullptr;
    *out_total = j->out_off[n_units];
    *job_out = j;
    return ZGPU_OK;
}

int zgpu_deflate_batch_finish(zgpu_job *j, uint8_t *out, uint64_t out_cap, uint64_t *out_off, uint32_t *crc32,
                              uint32_t *adler32) {
    if (!j) return ZGPU_E_ARG;
    int rc = ZGPU_OK;
    const size_t n = j->out_off.size() - 1;
    const uint64_t total = j->out_off[n];
    if (!out_off || (total && !out) || (crc32 && j->crc.empty()) || (adler32 && j->adler.empty())) {
        rc = ZGPU_E_ARG;
    } else if (total > out_cap) {
        rc = ZGPU_E_OUTCAP;
    } else {
        int cur = 0;
        const bool moved = hipGetDevice(&cur) == hipSuccess && cur != j->device && hipSetDevice(j->device) == hipSuccess;
        hipError_t e = total ? hipMemcpy(out, j->d_out, total, hipMemcpyDeviceToHost) : hipSuccess;
        if (moved) (void)hipSetDevice(cur);
        if (e != hipSuccess) {
            g_last_hip = e;
            rc = ZGPU_E_NODEVICE;
        } else {
            memcpy(out_off, j->out_off.data(), (n + 1) * sizeof(uint64_t));
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
    const bool moved = j->device >= 0 && hipGetDevice(&cur) == hipSuccess && cur != j->device &&
                       hipSetDevice(j->device) == hipSuccess;
    zgpu_plan_destroy(j->plan);
    (void)hipFree(j->d_in);
    (void)hipFree(j->d_out);
    if (moved) (void)hipSetDevice(cur);
    delete j;
}

int zgpu_deflate_batch(const uint8_t *in, const uint64_t *in_off, uint32_t n_units, int ctype, int lazy,
                       uint32_t flags, uint8_t *out, uint64_t out_cap, uint64_t *out_off,
                       uint32_t *crc32, uint32_t *adler32) {
    if (!in_off || !out || !out_off) return ZGPU_E_ARG;
    if (crc32) flags |= ZGPU_F_WANT_CRC;
    if (adler32) flags |= ZGPU_F_WANT_ADLER;
    zgpu_job *job = nullptr;
    uint64_t total = 0;
    int rc = zgpu_deflate_batch_begin(in, in_off, n_units, ctype, lazy, flags, &job, &total);
    if (rc) return rc;
    return zgpu_deflate_batch_finish(job, out, out_cap, out_off, crc32, adler32);
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
    DeviceState *ds = nullptr;
    int rc = get_state(&ds);
    if (rc) return rc;
    std::lock_guard<std::mutex> ck_lock(ds->ck_mu);
    hipStream_t st = (hipStream_t)stream_;
    // large buffers go through the wide kernel (one workgroup per CU, 4 MiB segments)
    const bool wide = len >= ZGPU_CKW_MIN_BYTES;
    const uint64_t seg = wide ? ZGPU_CKW_SEG : ZGPU_CK_SEG;
    const uint64_t nseg = len ? (len + seg - 1) / seg : 1;
    if (nseg + 1 > ds->seg_cap) {
        (void)hipFree(ds->d_seg_off);
        (void)hipFree(ds->d_seg_crc);
        (void)hipFree(ds->d_seg_adler);
        ds->seg_cap = 0;
        HIPCHK(hipMalloc(&ds->d_seg_off, (nseg + 1) * 8));
        HIPCHK(hipMalloc(&ds->d_seg_crc, nseg * 4));
        HIPCHK(hipMalloc(&ds->d_seg_adler, nseg * 4));
        ds->seg_cap = nseg + 1;
    }
    std::vector<uint64_t> off(nseg + 1);
    for (uint64_t i = 0; i < nseg; ++i) off[i] = i * seg;
    off[nseg] = len;
    HIPCHK(hipMemcpyAsync(ds->d_seg_off, off.data(), (nseg + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipStreamSynchronize(st));  // off is a local
    HIPCHK(hipEventRecord(ds->ev0, st));
    if (wide) {
        const int grid = (int)(nseg < (uint64_t)ds->cus ? nseg : (uint64_t)ds->cus);
        hipLaunchKernelGGL(zgpu_checksum_wide_kernel, dim3(grid), dim3(ZGPU_CKW_THREADS), 0, st,
                           (const uint8_t *)d_in, ds->d_seg_off, (uint32_t)nseg, ds->d_tables_wide,
                           ds->d_seg_crc, ds->d_seg_adler);
    } else {
        const int grid = (int)(nseg < (uint64_t)ds->cus * 8 ? nseg : (uint64_t)ds->cus * 8);
        hipLaunchKernelGGL(zgpu_checksum_kernel, dim3(grid), dim3(ZGPU_CK_THREADS), 0, st,
                           (const uint8_t *)d_in, ds->d_seg_off, (uint32_t)nseg, ds->d_tables,
                           ds->d_seg_crc, ds->d_seg_adler);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ds->ev1, st));
    std::vector<uint32_t> c(nseg), a(nseg);
    HIPCHK(hipMemcpyAsync(c.data(), ds->d_seg_crc, nseg * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(a.data(), ds->d_seg_adler, nseg * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipEventElapsedTime(&ds->last_ck_ms, ds->ev0, ds->ev1));
    uint32_t crc = c[0], ad = a[0];
    const uint32_t xs = h_xpow_bytes(seg);
    for (uint64_t i = 1; i < nseg; ++i) {
        const uint64_t l = off[i + 1] - off[i];
        crc = h_gf_mul(crc, l == seg ? xs : h_xpow_bytes(l)) ^ c[i];
        ad = zgpu_adler32_combine(ad, a[i], l);
    }
    if (crc32) *crc32 = crc;
    if (adler32) *adler32 = ad;
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
    HIPCHK(hipMemcpy(ptr + kNaTables, t.data(), t.size() * 4, hipMemcpyHostToDevice));
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
    uint8_t *d = nullptr;
    HIPCHK(hipMalloc(&d, len + 64));
    hipError_t e = len ? hipMemcpy(d, in, len, hipMemcpyHostToDevice) : hipSuccess;
    if (e != hipSuccess) {
        g_last_hip = e;
        (void)hipFree(d);
        return ZGPU_E_NODEVICE;
    }
    int rc = zgpu_checksum_device(d, len, crc32, adler32, nullptr);
    (void)hipFree(d);
    return rc;
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
