// Node.js N-API addon: binds the C ABI of libzgpu.so (include/zgpu.h) for the
// JavaScript shim in zlib_gpu.js.  N-API version <= 6 (builds against the
// Node 12 headers).  The addon holds no compression logic: it marshals typed
// arrays to zgpu_deflate_batch / zgpu_checksum and back.  libzgpu.so is loaded
// with dlopen at init(), so a missing library or GPU surfaces as a thrown
// Error, never as a silent fallback.
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <sys/mman.h>
#include <time.h>

#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

typedef int (*fn_deflate_batch)(const uint8_t *, const uint64_t *, uint32_t, int, int, uint32_t, uint8_t *,
                                uint64_t, uint64_t *, uint32_t *, uint32_t *);
typedef int (*fn_batch_begin)(const uint8_t *, const uint64_t *, uint32_t, int, int, uint32_t, void **, uint64_t *);
typedef int (*fn_batch_finish)(void *, uint8_t *, uint64_t, uint64_t *, uint32_t *, uint32_t *);
typedef int (*fn_batch_finish_spaced)(void *, uint8_t *, uint64_t, const uint32_t *, uint64_t *, uint32_t *, uint32_t *);
typedef int (*fn_checksum)(const uint8_t *, uint64_t, uint32_t *, uint32_t *);
typedef uint64_t (*fn_bound)(uint64_t);
typedef uint32_t (*fn_combine)(uint32_t, uint32_t, uint64_t);
typedef const char *(*fn_strerror)(int);
typedef int (*fn_set2)(int, int);
typedef int (*fn_set1)(int);

struct Lib {
    void *h = nullptr;
    fn_deflate_batch deflate_batch = nullptr;
    fn_batch_begin batch_begin = nullptr;
    fn_batch_finish batch_finish = nullptr;
    fn_batch_finish_spaced batch_finish_spaced = nullptr;
    void (*job_release)(void *) = nullptr;
    fn_checksum checksum = nullptr;
    fn_bound bound = nullptr;
    fn_combine crc_combine = nullptr, adler_combine = nullptr;
    fn_strerror strerror_ = nullptr;
    fn_set2 set_search = nullptr;
    fn_set1 set_min_lazy = nullptr;
    fn_set1 set_dense_search = nullptr;
    fn_set2 set_long_match = nullptr;
    int (*init)(const int *, int) = nullptr;
    int (*device_count)() = nullptr;
    int (*host_devices)(int *, int) = nullptr;
    int (*set_split)(uint64_t) = nullptr;
    void *(*host_alloc)(uint64_t) = nullptr;
    void (*host_free)(void *) = nullptr;
    void (*host_trim)() = nullptr;
} g;

#define NAPI_OK(call)                                            \
    do {                                                         \
        if ((call) != napi_ok) {                                 \
            napi_throw_error(env, nullptr, "N-API call failed: " #call); \
            return nullptr;                                      \
        }                                                        \
    } while (0)

// Output buffers of 4 MiB and more are mapped by the addon and asked to be backed by huge pages
// (MADV_HUGEPAGE), then handed to JavaScript as an external ArrayBuffer: a fresh 400 MB ArrayBuffer
// from V8's allocator costs about 100,000 page faults on first touch (88 ms on one thread, measured
// on the box: more than compressing the gigabyte it came from); in 2 MiB pages it is 200.  The
// mapping is released when the ArrayBuffer is collected.  Smaller buffers stay with V8's allocator.
struct Mapping {
    void *base;
    size_t len;
    int64_t accounted;
};
void unmap_output(napi_env env, void *, void *hint) {
    Mapping *m = (Mapping *)hint;
    munmap(m->base, m->len);
    int64_t now = 0;
    napi_adjust_external_memory(env, -m->accounted, &now);
    delete m;
}
// An anonymous mapping of `room` bytes starting on a 2 MiB boundary (pages are committed when first written:
// a mapping sized for the worst case costs address space only); null if it cannot be had.
void *map_output(uint64_t room, void **base, size_t *len) {
    const size_t kHuge = 2u << 20;
    *len = ((size_t)room + kHuge - 1) / kHuge * kHuge + kHuge;  // room to start on a 2 MiB boundary
    *base = mmap(nullptr, *len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (*base == MAP_FAILED) return nullptr;
    void *start = (void *)(((uintptr_t)*base + kHuge - 1) / kHuge * kHuge);
    (void)madvise(start, *len - ((uintptr_t)start - (uintptr_t)*base), MADV_HUGEPAGE);  // advice only
    return start;
}
// hands the first `used` bytes of such a mapping to JavaScript; the whole mapping goes when the ArrayBuffer is collected
bool adopt_output(napi_env env, void *base, size_t len, void *start, uint64_t used, napi_value *ab) {
    Mapping *m = new (std::nothrow) Mapping{base, len, (int64_t)used};
    if (m && napi_create_external_arraybuffer(env, start, (size_t)used, unmap_output, m, ab) == napi_ok) {
        int64_t now = 0;
        napi_adjust_external_memory(env, (int64_t)used, &now);  // what the collector should weigh the buffer at
        return true;
    }
    delete m;
    munmap(base, len);
    return false;
}
bool make_output_buffer(napi_env env, uint64_t room, void **data, napi_value *ab) {
    const size_t kHuge = 2u << 20;
    if (room >= (4u << 20) && !getenv("ZGPU_NO_HUGE_OUTPUT")) {
        const size_t len = ((size_t)room + kHuge - 1) / kHuge * kHuge + kHuge;  // room to start on a 2 MiB boundary
        void *base = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (base != MAP_FAILED) {
            void *start = (void *)(((uintptr_t)base + kHuge - 1) / kHuge * kHuge);
            (void)madvise(start, len - ((uintptr_t)start - (uintptr_t)base), MADV_HUGEPAGE);  // advice only
            Mapping *m = new (std::nothrow) Mapping{base, len, (int64_t)room};
            if (m && napi_create_external_arraybuffer(env, start, (size_t)room, unmap_output, m, ab) == napi_ok) {
                int64_t now = 0;
                napi_adjust_external_memory(env, (int64_t)room, &now);  // what the collector should weigh the buffer at
                *data = start;
                return true;
            }
            delete m;
            munmap(base, len);
        }
    }
    return napi_create_arraybuffer(env, (size_t)room, data, ab) == napi_ok;
}

// Inputs that do not lie back to back are gathered here before they go to the library.  One mapping
// for the life of the process, grown when a call needs more; a call holds g_gather_mu while it uses it.
std::mutex g_gather_mu;  // held by a call for as long as it reads the gathering buffer (worker threads may load the addon too)
uint8_t *gather_buffer(size_t need) {
    static uint8_t *buf = nullptr;
    static size_t cap = 0;
    if (need <= cap) return buf;
    const size_t kHuge = 2u << 20;
    const size_t len = (need + need / 4 + kHuge - 1) / kHuge * kHuge;
    void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) return nullptr;
    (void)madvise(p, len, MADV_HUGEPAGE);
    if (buf) munmap(buf, cap);
    buf = (uint8_t *)p;
    cap = len;
    return buf;
}

napi_value throw_zgpu(napi_env env, const char *what, int rc) {
    std::string msg = std::string(what) + ": " + (g.strerror_ ? g.strerror_(rc) : "libzgpu error") +
                      " (code " + std::to_string(rc) + ")";
    napi_throw_error(env, nullptr, msg.c_str());
    return nullptr;
}

bool need_lib(napi_env env) {
    if (g.h) return true;
    napi_throw_error(env, nullptr, "libzgpu.so is not loaded: call init(path) first (there is no CPU fallback)");
    return false;
}

// Names the devices the batch calls may share: "all", or a list "0,2,3".  Touches the HIP runtime.
napi_value name_devices(napi_env env, const std::string &spec) {
    std::vector<int> ids;
    if (spec == "all") {
        const int n = g.device_count();
        if (n < 0) return throw_zgpu(env, "zgpu_device_count", n);
        for (int i = 0; i < n; ++i) ids.push_back(i);
    } else {
        size_t p = 0;
        while (p < spec.size()) {
            size_t q = spec.find(',', p);
            if (q == std::string::npos) q = spec.size();
            const std::string tok = spec.substr(p, q - p);
            if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) {
                napi_throw_type_error(env, nullptr, "devices: \"all\" or a comma-separated list of device numbers");
                return nullptr;
            }
            // all digits, but possibly more of them than an int holds
            char *tail = nullptr;
            const unsigned long v = strtoul(tok.c_str(), &tail, 10);
            if (tok.size() > 9 || v > 0x7fffffffUL) {
                napi_throw_type_error(env, nullptr, "devices: device number out of range");
                return nullptr;
            }
            ids.push_back((int)v);
            p = q + 1;
        }
    }
    const int rc = g.init(ids.data(), (int)ids.size());
    if (rc != 0) return throw_zgpu(env, "zgpu_init", rc);
    return nullptr;
}

// init(pathToLibzgpu[, devices]): loads the library; devices ("all" or "0,1,...") names the GPUs a batch is
// split over (zgpu_init).  Without it the calls run on the current device and nothing touches the GPU here.
napi_value Init_(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    char path[4096];
    size_t len = 0;
    NAPI_OK(napi_get_value_string_utf8(env, argv[0], path, sizeof path, &len));
    std::string devices;
    if (argc >= 2) {
        napi_valuetype vt;
        NAPI_OK(napi_typeof(env, argv[1], &vt));
        if (vt == napi_string) {
            char buf[256];
            size_t dl = 0;
            NAPI_OK(napi_get_value_string_utf8(env, argv[1], buf, sizeof buf, &dl));
            devices.assign(buf, dl);
        }
    }
    if (g.h) return devices.empty() ? nullptr : name_devices(env, devices);
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        napi_throw_error(env, nullptr, (std::string("cannot load ") + path + ": " + dlerror()).c_str());
        return nullptr;
    }
    g.deflate_batch = (fn_deflate_batch)dlsym(h, "zgpu_deflate_batch");
    g.batch_begin = (fn_batch_begin)dlsym(h, "zgpu_deflate_batch_begin");
    g.batch_finish = (fn_batch_finish)dlsym(h, "zgpu_deflate_batch_finish");
    g.batch_finish_spaced = (fn_batch_finish_spaced)dlsym(h, "zgpu_deflate_batch_finish_spaced");
    g.job_release = (void (*)(void *))dlsym(h, "zgpu_job_release");
    g.checksum = (fn_checksum)dlsym(h, "zgpu_checksum");
    g.bound = (fn_bound)dlsym(h, "zgpu_deflate_bound");
    g.crc_combine = (fn_combine)dlsym(h, "zgpu_crc32_combine");
    g.adler_combine = (fn_combine)dlsym(h, "zgpu_adler32_combine");
    g.strerror_ = (fn_strerror)dlsym(h, "zgpu_strerror");
    g.set_search = (fn_set2)dlsym(h, "zgpu_set_search");
    g.set_min_lazy = (fn_set1)dlsym(h, "zgpu_set_min_lazy");
    g.set_dense_search = (fn_set1)dlsym(h, "zgpu_set_dense_search");
    g.set_long_match = (fn_set2)dlsym(h, "zgpu_set_long_match");
    g.set_split = (int (*)(uint64_t))dlsym(h, "zgpu_set_split");
    g.init = (int (*)(const int *, int))dlsym(h, "zgpu_init");
    g.device_count = (int (*)())dlsym(h, "zgpu_device_count");
    g.host_devices = (int (*)(int *, int))dlsym(h, "zgpu_host_devices");
    g.host_alloc = (void *(*)(uint64_t))dlsym(h, "zgpu_host_alloc");
    g.host_free = (void (*)(void *))dlsym(h, "zgpu_host_free");
    g.host_trim = (void (*)())dlsym(h, "zgpu_host_trim");
    if (!g.deflate_batch || !g.batch_begin || !g.batch_finish || !g.batch_finish_spaced || !g.job_release || !g.checksum || !g.bound || !g.crc_combine || !g.adler_combine || !g.strerror_ ||
        !g.set_search || !g.set_min_lazy || !g.set_dense_search || !g.set_split || !g.set_long_match || !g.init || !g.device_count || !g.host_devices ||
        !g.host_alloc || !g.host_free || !g.host_trim) {
        dlclose(h);
        napi_throw_error(env, nullptr, "libzgpu.so lacks an expected symbol");
        return nullptr;
    }
    g.h = h;
    return devices.empty() ? nullptr : name_devices(env, devices);
}

// hostDevices() -> the device numbers a large batch would be split over
napi_value HostDevices(napi_env env, napi_callback_info) {
    if (!need_lib(env)) return nullptr;
    int ids[64];
    const int n = g.host_devices(ids, 64);
    if (n < 0) return throw_zgpu(env, "zgpu_host_devices", n);
    napi_value arr;
    NAPI_OK(napi_create_array_with_length(env, (size_t)n, &arr));
    for (int i = 0; i < n && i < 64; ++i) {
        napi_value v;
        NAPI_OK(napi_create_int32(env, ids[i], &v));
        NAPI_OK(napi_set_element(env, arr, (uint32_t)i, v));
    }
    return arr;
}

bool get_u8(napi_env env, napi_value v, const uint8_t **data, size_t *len) {
    bool is_ta = false;
    if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta) return false;
    napi_typedarray_type t;
    void *p = nullptr;
    napi_value ab;
    size_t off = 0;
    if (napi_get_typedarray_info(env, v, &t, len, &p, &ab, &off) != napi_ok) return false;
    if (t != napi_uint8_array && t != napi_uint8_clamped_array && t != napi_int8_array) return false;
    *data = (const uint8_t *)p;
    return true;
}

// deflateBatch(inputs: Uint8Array[], ctype, lazy, flags, wantCrc, wantAdler)
//   -> { outputs: Uint8Array[] (views into one ArrayBuffer), crc32: Uint32Array|null, adler32: Uint32Array|null }
napi_value DeflateBatch(napi_env env, napi_callback_info info) {
    if (!need_lib(env)) return nullptr;
    size_t argc = 8;
    napi_value argv[8];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    if (argc < 6) {
        napi_throw_type_error(env, nullptr, "deflateBatch(inputs, ctype, lazy, flags, wantCrc, wantAdler)");
        return nullptr;
    }
    uint32_t n = 0;
    NAPI_OK(napi_get_array_length(env, argv[0], &n));
    int32_t ctype = 0, lazy = 0;
    uint32_t flags = 0;
    bool want_crc = false, want_adler = false;
    NAPI_OK(napi_get_value_int32(env, argv[1], &ctype));
    NAPI_OK(napi_get_value_int32(env, argv[2], &lazy));
    NAPI_OK(napi_get_value_uint32(env, argv[3], &flags));
    NAPI_OK(napi_get_value_bool(env, argv[4], &want_crc));
    NAPI_OK(napi_get_value_bool(env, argv[5], &want_adler));
    napi_value result;
    NAPI_OK(napi_create_object(env, &result));
    napi_value outs;
    NAPI_OK(napi_create_array_with_length(env, n, &outs));
    napi_value null_v;
    NAPI_OK(napi_get_null(env, &null_v));
    napi_value crc_v = null_v, adler_v = null_v, buffer_v = null_v;
    // spaced mode (for callers that frame every unit themselves, like Zip#compress()): gaps[i] bytes are
    // left free in front of unit i's stream and `tail` bytes behind the last; the result also carries the
    // one ArrayBuffer (`buffer`) and the end of the last stream (`end`)
    const uint32_t *gaps = nullptr;
    double tail = 0;
    uint8_t *own_out = nullptr;
    size_t own_cap = 0, own_at = 0;
    napi_value own_ab = nullptr;
    if (argc >= 7) {
        bool is_ta = false;
        if (napi_is_typedarray(env, argv[6], &is_ta) == napi_ok && is_ta) {
            napi_typedarray_type tt;
            size_t gl = 0, goff = 0;
            void *gp = nullptr;
            napi_value gab;
            NAPI_OK(napi_get_typedarray_info(env, argv[6], &tt, &gl, &gp, &gab, &goff));
            if (tt == napi_uint8_array) {
                // the caller's own output buffer (kept and reused from call to call; page-locked if it came
                // from allocInput): the streams are written into it back to back and the outputs are views of it
                own_out = (uint8_t *)gp;
                own_cap = gl;
                own_ab = gab;
                own_at = goff;
            } else {
                if (tt != napi_uint32_array || gl != n) {
                    napi_throw_type_error(env, nullptr, "gaps: a Uint32Array with one entry per input (or a Uint8Array to write into)");
                    return nullptr;
                }
                gaps = (const uint32_t *)gp;
                if (argc >= 8) NAPI_OK(napi_get_value_double(env, argv[7], &tail));
                if (tail < 0 || tail > 4294967295.0) tail = 0;
            }
        }
    }
    double end_pos = 0;
    static const bool timing = getenv("ZGPU_ADDON_TIMING") != nullptr;  // diagnostic: where a call's time goes
    auto now_ms = [] {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    };
    const double t_start = timing ? now_ms() : 0;
    double t_in = 0, t_call = 0;
    if (n) {
        std::vector<uint64_t> in_off(n + 1, 0), out_off(n + 1, 0);
        std::vector<const uint8_t *> ptrs(n);
        for (uint32_t i = 0; i < n; ++i) {
            napi_value e;
            NAPI_OK(napi_get_element(env, argv[0], i, &e));
            size_t len = 0;
            if (!get_u8(env, e, &ptrs[i], &len)) {
                napi_throw_type_error(env, nullptr, "inputs must be Uint8Array");
                return nullptr;
            }
            in_off[i + 1] = in_off[i] + len;
        }
        // Inputs that already lie back to back in memory (one input; subarrays of one buffer,
        // as the chunked helpers pass) are handed over in place; anything else is gathered
        // into one staging buffer.  Neither buffer is cleared first.
        const uint8_t *first = nullptr;  // where input 0 would start if all of them lie back to back
        bool contiguous = true;
        for (uint32_t i = 0; i < n && contiguous; ++i) {
            if (in_off[i + 1] == in_off[i]) continue;  // an empty input has no address to check
            if (!first) first = ptrs[i] - in_off[i];
            contiguous = ptrs[i] == first + in_off[i];
        }
        const uint8_t *src = first;
        std::unique_lock<std::mutex> gather_lock(g_gather_mu, std::defer_lock);
        if (!contiguous || !first) {
            gather_lock.lock();
            // the gathering buffer is kept between calls (it only grows): pages touched once stay
            uint8_t *flat = gather_buffer(in_off[n] ? in_off[n] : 1);
            if (!flat) return throw_zgpu(env, "staging buffer", -3);
            for (uint32_t i = 0; i < n; ++i)
                if (in_off[i + 1] > in_off[i]) memcpy(flat + in_off[i], ptrs[i], in_off[i + 1] - in_off[i]);
            src = flat;
        }
        std::vector<uint32_t> crc(want_crc ? n : 0), adler(want_adler ? n : 0);
        int rc = 0;
        if (timing) t_in = now_ms();
        // Large batches, one step: the library writes the streams straight into a mapping sized for the worst
        // case (address space only: pages are committed as the streams land), whose first `total` bytes then
        // become the ArrayBuffer -- no staging of the result in page-locked memory and no second copy of it
        // (1 GiB: 17 -> 25-30 GB/s through Node).
        bool one_step_done = false;
        if (own_out) {
            // the streams are written while later inputs are still being read: the two must not share memory
            if (src < own_out + own_cap && own_out < src + in_off[n]) {
                napi_throw_type_error(env, nullptr, "output: the buffer overlaps the inputs");
                return nullptr;
            }
            rc = g.deflate_batch(src, in_off.data(), n, ctype, lazy, flags, own_out, own_cap, out_off.data(),
                                 want_crc ? crc.data() : nullptr, want_adler ? adler.data() : nullptr);
            if (rc) return throw_zgpu(env, "zgpu_deflate_batch", rc);
            if (timing) t_call = now_ms();
            for (uint32_t i = 0; i < n; ++i) {
                napi_value ta;
                NAPI_OK(napi_create_typedarray(env, napi_uint8_array, out_off[i + 1] - out_off[i], own_ab, own_at + out_off[i], &ta));
                NAPI_OK(napi_set_element(env, outs, i, ta));
            }
            one_step_done = true;
        } else if (!gaps && in_off[n] >= (32u << 20) && !getenv("ZGPU_NO_HUGE_OUTPUT")) {
            uint64_t bound = 64;
            for (uint32_t i = 0; i < n; ++i) bound += g.bound(in_off[i + 1] - in_off[i]) + 48;
            void *base = nullptr;
            size_t maplen = 0;
            void *start = map_output(bound, &base, &maplen);
            if (start) {
                // The mapping's pages come into being when first written, each zeroed by the kernel: for the 0.4 GB
                // that a gigabyte compresses to, more time than compressing it takes if it all happens on the
                // threads that copy the streams down.  Two helper threads touch the front of the mapping (half the
                // input's size: more than most data compresses to) while the GPU works; what they touched beyond
                // the result is given back below.
                uint64_t pre = in_off[n] / 2 < bound ? in_off[n] / 2 : bound;
                pre &= ~(uint64_t)4095;
                std::thread toucher[2];
                for (int t = 0; t < 2; ++t)
                    toucher[t] = std::thread([=] {
                        // an atomic OR of nothing: a write access, so the page comes into being, that cannot
                        // undo a byte the library has stored there in the meantime
                        uint8_t *p = (uint8_t *)start;
                        for (uint64_t o = (pre / 2 * t) & ~(uint64_t)4095, e = t ? pre : pre / 2; o < e; o += 4096)
                            (void)__atomic_fetch_or(p + o, (uint8_t)0, __ATOMIC_RELAXED);
                    });
                rc = g.deflate_batch(src, in_off.data(), n, ctype, lazy, flags, (uint8_t *)start, bound, out_off.data(),
                                     want_crc ? crc.data() : nullptr, want_adler ? adler.data() : nullptr);
                for (auto &t : toucher) t.join();
                if (rc == 0 && pre > out_off[n]) {
                    const uint64_t keep = (out_off[n] + 4095) & ~(uint64_t)4095;
                    if (pre > keep) (void)madvise((uint8_t *)start + keep, pre - keep, MADV_DONTNEED);
                }
                if (rc) {
                    munmap(base, maplen);
                    return throw_zgpu(env, "zgpu_deflate_batch", rc);
                }
                if (timing) t_call = now_ms();
                napi_value ab;
                if (!adopt_output(env, base, maplen, start, out_off[n], &ab)) {
                    napi_throw_error(env, nullptr, "cannot create the output ArrayBuffer");
                    return nullptr;
                }
                for (uint32_t i = 0; i < n; ++i) {
                    napi_value ta;
                    NAPI_OK(napi_create_typedarray(env, napi_uint8_array, out_off[i + 1] - out_off[i], ab, out_off[i], &ta));
                    NAPI_OK(napi_set_element(env, outs, i, ta));
                }
                one_step_done = true;
            }
        }
        if (!one_step_done) {
        // Two steps: compress, learn the size, then let the library copy into an ArrayBuffer of exactly
        // that size (see make_output_buffer).
        void *job = nullptr;
        uint64_t total = 0;
        rc = g.batch_begin(src, in_off.data(), n, ctype, lazy,
                               flags | (want_crc ? 16u : 0u) | (want_adler ? 32u : 0u), &job, &total);
        if (rc) return throw_zgpu(env, "zgpu_deflate_batch_begin", rc);
        // from here to batch_finish (which releases the job itself, also when it fails) every way out
        // releases the job: the compressed bytes it holds live in device memory
        struct JobGuard {
            void *job;
            ~JobGuard() {
                if (job) g.job_release(job);
            }
        } guard{job};
        uint64_t room = total;
        if (gaps) {
            for (uint32_t i = 0; i < n; ++i) room += gaps[i];
            room += (uint64_t)tail;
        }
        void *ab_data = nullptr;
        napi_value ab;
        if (!make_output_buffer(env, room, &ab_data, &ab)) {
            napi_throw_error(env, nullptr, "cannot allocate the output ArrayBuffer");
            return nullptr;
        }
        guard.job = nullptr;
        if (gaps)
            rc = g.batch_finish_spaced(job, (uint8_t *)ab_data, room, gaps, out_off.data(), want_crc ? crc.data() : nullptr,
                                       want_adler ? adler.data() : nullptr);
        else
            rc = g.batch_finish(job, (uint8_t *)ab_data, total, out_off.data(), want_crc ? crc.data() : nullptr,
                                want_adler ? adler.data() : nullptr);
        if (rc) return throw_zgpu(env, "zgpu_deflate_batch_finish", rc);
        if (gaps) {
            // positions of a spaced result: unit i's stream ends where the gap of unit i + 1 begins
            buffer_v = ab;
            end_pos = (double)out_off[n];
            for (uint32_t i = 0; i < n; ++i) {
                const uint64_t stop = i + 1 < n ? out_off[i + 1] - gaps[i + 1] : out_off[n];
                napi_value ta;
                NAPI_OK(napi_create_typedarray(env, napi_uint8_array, stop - out_off[i], ab, out_off[i], &ta));
                NAPI_OK(napi_set_element(env, outs, i, ta));
            }
        } else
        for (uint32_t i = 0; i < n; ++i) {
            napi_value ta;
            NAPI_OK(napi_create_typedarray(env, napi_uint8_array, out_off[i + 1] - out_off[i], ab, out_off[i], &ta));
            NAPI_OK(napi_set_element(env, outs, i, ta));
        }
        }
        auto make_u32 = [&](const std::vector<uint32_t> &v, napi_value *dst) -> bool {
            void *d = nullptr;
            napi_value b;
            if (napi_create_arraybuffer(env, v.size() * 4, &d, &b) != napi_ok) return false;
            memcpy(d, v.data(), v.size() * 4);
            return napi_create_typedarray(env, napi_uint32_array, v.size(), b, 0, dst) == napi_ok;
        };
        if (want_crc && !make_u32(crc, &crc_v)) return nullptr;
        if (want_adler && !make_u32(adler, &adler_v)) return nullptr;
    }
    if (timing && t_call > 0)
        fprintf(stderr, "deflateBatch: inputs %.2f ms, library call %.2f ms, outputs %.2f ms\n", t_in - t_start, t_call - t_in,
                now_ms() - t_call);
    NAPI_OK(napi_set_named_property(env, result, "outputs", outs));
    NAPI_OK(napi_set_named_property(env, result, "crc32", crc_v));
    NAPI_OK(napi_set_named_property(env, result, "adler32", adler_v));
    if (gaps) {
        napi_value end_v;
        NAPI_OK(napi_create_double(env, end_pos, &end_v));
        NAPI_OK(napi_set_named_property(env, result, "buffer", buffer_v));
        NAPI_OK(napi_set_named_property(env, result, "end", end_v));
    }
    return result;
}

// checksum(u8) -> [crc32, adler32]
napi_value Checksum(napi_env env, napi_callback_info info) {
    if (!need_lib(env)) return nullptr;
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    const uint8_t *p = nullptr;
    size_t len = 0;
    if (argc < 1 || !get_u8(env, argv[0], &p, &len)) {
        napi_throw_type_error(env, nullptr, "checksum(Uint8Array)");
        return nullptr;
    }
    uint32_t c = 0, a = 1;
    int rc = g.checksum(p, len, &c, &a);
    if (rc) return throw_zgpu(env, "zgpu_checksum", rc);
    napi_value arr, v0, v1;
    NAPI_OK(napi_create_array_with_length(env, 2, &arr));
    NAPI_OK(napi_create_uint32(env, c, &v0));
    NAPI_OK(napi_create_uint32(env, a, &v1));
    NAPI_OK(napi_set_element(env, arr, 0, v0));
    NAPI_OK(napi_set_element(env, arr, 1, v1));
    return arr;
}

napi_value combine(napi_env env, napi_callback_info info, bool crc) {
    if (!need_lib(env)) return nullptr;
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    uint32_t a = 0, b = 0;
    double len = 0;
    NAPI_OK(napi_get_value_uint32(env, argv[0], &a));
    NAPI_OK(napi_get_value_uint32(env, argv[1], &b));
    NAPI_OK(napi_get_value_double(env, argv[2], &len));
    uint32_t r = (crc ? g.crc_combine : g.adler_combine)(a, b, (uint64_t)len);
    napi_value v;
    NAPI_OK(napi_create_uint32(env, r, &v));
    return v;
}
napi_value Crc32Combine(napi_env env, napi_callback_info info) { return combine(env, info, true); }
napi_value Adler32Combine(napi_env env, napi_callback_info info) { return combine(env, info, false); }

napi_value DeflateBound(napi_env env, napi_callback_info info) {
    if (!need_lib(env)) return nullptr;
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    double n = 0;
    NAPI_OK(napi_get_value_double(env, argv[0], &n));
    napi_value v;
    NAPI_OK(napi_create_double(env, (double)g.bound((uint64_t)n), &v));
    return v;
}

napi_value SetSearch(napi_env env, napi_callback_info info) {
    if (!need_lib(env)) return nullptr;
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    int32_t a = 0, b = 32;
    NAPI_OK(napi_get_value_int32(env, argv[0], &a));
    if (argc > 1) NAPI_OK(napi_get_value_int32(env, argv[1], &b));
    int rc = g.set_search(a, b);
    if (rc) return throw_zgpu(env, "zgpu_set_search", rc);
    return nullptr;
}

napi_value SetLongMatch(napi_env env, napi_callback_info info) {
    if (!need_lib(env)) return nullptr;
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    int32_t a = 0, b = 1;
    if (argc > 0) NAPI_OK(napi_get_value_int32(env, argv[0], &a));
    if (argc > 1) NAPI_OK(napi_get_value_int32(env, argv[1], &b));
    int rc = g.set_long_match(a, b);
    if (rc) return throw_zgpu(env, "zgpu_set_long_match", rc);
    return nullptr;
}

napi_value SetMinLazy(napi_env env, napi_callback_info info) {
    if (!need_lib(env)) return nullptr;
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    int32_t a = 0;
    NAPI_OK(napi_get_value_int32(env, argv[0], &a));
    int rc = g.set_min_lazy(a);
    if (rc) return throw_zgpu(env, "zgpu_set_min_lazy", rc);
    return nullptr;
}

napi_value SetDenseSearch(napi_env env, napi_callback_info info) {
    if (!need_lib(env)) return nullptr;
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    int32_t a = 0;
    NAPI_OK(napi_get_value_int32(env, argv[0], &a));
    int rc = g.set_dense_search(a);
    if (rc) return throw_zgpu(env, "zgpu_set_dense_search", rc);
    return nullptr;
}

napi_value SetSplit(napi_env env, napi_callback_info info) {
    if (!need_lib(env)) return nullptr;
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    double a = 0;
    NAPI_OK(napi_get_value_double(env, argv[0], &a));
    if (!(a >= 0 && a <= 9007199254740992.0)) return throw_zgpu(env, "zgpu_set_split", -2);
    int rc = g.set_split((uint64_t)a);
    if (rc) return throw_zgpu(env, "zgpu_set_split", rc);
    return nullptr;
}

// allocInput(bytes) -> ArrayBuffer over page-locked memory (zgpu_host_alloc): inputs that live in it go to the
// device by DMA at the full PCIe rate instead of through the runtime's staging; freed when collected
void free_locked(napi_env env, void *data, void *hint) {
    g.host_free(data);
    int64_t now = 0;
    napi_adjust_external_memory(env, -(int64_t)(intptr_t)hint, &now);
}
napi_value AllocInput(napi_env env, napi_callback_info info) {
    if (!need_lib(env)) return nullptr;
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    double bytes = 0;
    if (argc < 1 || napi_get_value_double(env, argv[0], &bytes) != napi_ok || !(bytes >= 0) || bytes > 1099511627776.0) {
        napi_throw_type_error(env, nullptr, "allocInput(bytes)");
        return nullptr;
    }
    void *p = g.host_alloc((uint64_t)bytes);
    if (!p) return throw_zgpu(env, "zgpu_host_alloc", -3);
    napi_value ab;
    if (napi_create_external_arraybuffer(env, p, (size_t)bytes, free_locked, (void *)(intptr_t)(int64_t)bytes, &ab) != napi_ok) {
        g.host_free(p);
        napi_throw_error(env, nullptr, "cannot create the ArrayBuffer");
        return nullptr;
    }
    int64_t now = 0;
    napi_adjust_external_memory(env, (int64_t)bytes, &now);
    return ab;
}

// hostTrim(): releases the staging buffers the library keeps between calls
napi_value HostTrim(napi_env env, napi_callback_info) {
    if (!need_lib(env)) return nullptr;
    g.host_trim();
    return nullptr;
}

napi_value ModuleInit(napi_env env, napi_value exports) {
    const napi_property_descriptor props[] = {
        {"init", nullptr, Init_, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"hostDevices", nullptr, HostDevices, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"deflateBatch", nullptr, DeflateBatch, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"checksum", nullptr, Checksum, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"crc32Combine", nullptr, Crc32Combine, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"adler32Combine", nullptr, Adler32Combine, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"deflateBound", nullptr, DeflateBound, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setSearch", nullptr, SetSearch, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setLongMatch", nullptr, SetLongMatch, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setMinLazy", nullptr, SetMinLazy, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setDenseSearch", nullptr, SetDenseSearch, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"setSplit", nullptr, SetSplit, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"allocInput", nullptr, AllocInput, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
        {"hostTrim", nullptr, HostTrim, nullptr, nullptr, nullptr, napi_enumerable, nullptr},
    };
    if (napi_define_properties(env, exports, sizeof props / sizeof props[0], props) != napi_ok) return nullptr;
    return exports;
}

}  // namespace

NAPI_MODULE(zgpu_addon, ModuleInit)
