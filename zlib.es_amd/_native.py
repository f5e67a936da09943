"""ctypes binding of libzgpu.so (the C ABI in include/zgpu.h).

The library is the only compute path: if it is missing or no GPU is usable the
calls raise, they never fall back to a CPU implementation.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ZGPU_LIB selects a diagnostic build (e.g. libzgpu_prof.so); the default is the product library
LIB_PATH = os.environ.get("ZGPU_LIB") or os.path.join(_HERE, "libzgpu.so")

NONE, FIXED, DYNAMIC = 0, 1, 2
F_FINAL_EACH, F_SYNC_ALIGN, F_FINAL_LAST, F_PRIME_WINDOW, F_WANT_CRC, F_WANT_ADLER, F_PRIME_FIRST = 1, 2, 4, 8, 16, 32, 64

# every symbol include/zgpu.h declares
SYMBOLS = [
    "zgpu_init", "zgpu_device_count", "zgpu_shutdown", "zgpu_host_devices", "zgpu_partition_units", "zgpu_strerror", "zgpu_last_hip_error", "zgpu_set_search", "zgpu_set_split",
    "zgpu_set_min_lazy", "zgpu_set_long_match", "zgpu_set_dense_search",
    "zgpu_deflate_bound", "zgpu_deflate_batch", "zgpu_deflate_batch_begin", "zgpu_deflate_batch_finish",
    "zgpu_deflate_batch_finish_spaced",
    "zgpu_job_release", "zgpu_plan_create", "zgpu_plan_destroy",
    "zgpu_plan_out_bound", "zgpu_plan_run", "zgpu_plan_out_off_device", "zgpu_plan_crc_device",
    "zgpu_plan_adler_device", "zgpu_plan_fetch", "zgpu_plan_kernel_ms", "zgpu_plan_phase_ms", "zgpu_checksum",
    "zgpu_checksum_device", "zgpu_noalloc_setup", "zgpu_noalloc_max_bytes",
    "zgpu_checksum_device_noalloc", "zgpu_crc32_update", "zgpu_adler32_update", "zgpu_checksum_kernel_ms",
    "zgpu_crc32_combine", "zgpu_adler32_combine", "zgpu_host_alloc", "zgpu_host_free", "zgpu_host_trim",
]

_lib = None


class ZgpuError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = lib().zgpu_strerror(code).decode() if _lib is not None else "libzgpu error"
        hip = lib().zgpu_last_hip_error() if _lib is not None else 0
        super().__init__("%s: %s (code %d, hip %d)" % (what or "zgpu", msg, code, hip))


def lib():
    """Load libzgpu.so (built in-tree by zlib.es_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libzgpu.so not built: run `make -C zlib.es_amd/csrc` "
                          "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
    L.zgpu_init.restype = i32
    L.zgpu_init.argtypes = [C.POINTER(i32), i32]
    L.zgpu_shutdown.restype = None
    L.zgpu_device_count.restype = i32
    L.zgpu_device_count.argtypes = []
    L.zgpu_host_devices.restype = i32
    L.zgpu_host_devices.argtypes = [vp, i32]
    L.zgpu_partition_units.restype = i32
    L.zgpu_partition_units.argtypes = [vp, u32, i32, vp]
    L.zgpu_strerror.restype = C.c_char_p
    L.zgpu_strerror.argtypes = [i32]
    L.zgpu_last_hip_error.restype = i32
    L.zgpu_set_search.restype = i32
    L.zgpu_set_search.argtypes = [i32, i32]
    L.zgpu_set_split.restype = i32
    L.zgpu_set_split.argtypes = [C.c_uint64]
    L.zgpu_set_min_lazy.restype = i32
    L.zgpu_set_min_lazy.argtypes = [i32]
    L.zgpu_set_dense_search.restype = i32
    L.zgpu_set_dense_search.argtypes = [i32]
    L.zgpu_set_long_match.restype = i32
    L.zgpu_set_long_match.argtypes = [i32, i32]
    L.zgpu_deflate_bound.restype = u64
    L.zgpu_deflate_bound.argtypes = [u64]
    L.zgpu_deflate_batch.restype = i32
    L.zgpu_deflate_batch.argtypes = [vp, vp, u32, i32, i32, u32, vp, u64, vp, vp, vp]
    L.zgpu_deflate_batch_begin.restype = i32
    L.zgpu_deflate_batch_begin.argtypes = [vp, vp, u32, i32, i32, u32, C.POINTER(vp), C.POINTER(u64)]
    L.zgpu_deflate_batch_finish.restype = i32
    L.zgpu_deflate_batch_finish.argtypes = [vp, vp, u64, vp, vp, vp]
    L.zgpu_deflate_batch_finish_spaced.restype = i32
    L.zgpu_deflate_batch_finish_spaced.argtypes = [vp, vp, u64, vp, vp, vp, vp]
    L.zgpu_job_release.restype = None
    L.zgpu_job_release.argtypes = [vp]
    L.zgpu_plan_create.restype = i32
    L.zgpu_plan_create.argtypes = [C.POINTER(vp), vp, u32, i32, i32, u32]
    L.zgpu_plan_destroy.restype = None
    L.zgpu_plan_destroy.argtypes = [vp]
    L.zgpu_plan_out_bound.restype = u64
    L.zgpu_plan_out_bound.argtypes = [vp]
    L.zgpu_plan_run.restype = i32
    L.zgpu_plan_run.argtypes = [vp, vp, vp, u64, vp]
    for n in ("zgpu_plan_out_off_device", "zgpu_plan_crc_device", "zgpu_plan_adler_device"):
        getattr(L, n).restype = vp
        getattr(L, n).argtypes = [vp]
    L.zgpu_plan_fetch.restype = i32
    L.zgpu_plan_fetch.argtypes = [vp, vp, vp, vp, vp]
    L.zgpu_plan_kernel_ms.restype = i32
    L.zgpu_plan_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.zgpu_plan_phase_ms.restype = i32
    L.zgpu_plan_phase_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.zgpu_checksum.restype = i32
    L.zgpu_checksum.argtypes = [vp, u64, C.POINTER(u32), C.POINTER(u32)]
    L.zgpu_checksum_device.restype = i32
    L.zgpu_checksum_device.argtypes = [vp, u64, C.POINTER(u32), C.POINTER(u32), vp]
    L.zgpu_noalloc_setup.restype = i32
    L.zgpu_noalloc_setup.argtypes = []
    L.zgpu_noalloc_max_bytes.restype = u64
    L.zgpu_noalloc_max_bytes.argtypes = []
    L.zgpu_checksum_device_noalloc.restype = i32
    L.zgpu_checksum_device_noalloc.argtypes = [vp, u64, vp, vp]
    L.zgpu_crc32_update.restype = i32
    L.zgpu_crc32_update.argtypes = [vp, u64, u32, C.POINTER(u32)]
    L.zgpu_adler32_update.restype = i32
    L.zgpu_adler32_update.argtypes = [vp, u64, u32, C.POINTER(u32)]
    L.zgpu_checksum_kernel_ms.restype = i32
    L.zgpu_checksum_kernel_ms.argtypes = [C.POINTER(C.c_float)]
    L.zgpu_crc32_combine.restype = u32
    L.zgpu_crc32_combine.argtypes = [u32, u32, u64]
    L.zgpu_adler32_combine.restype = u32
    L.zgpu_adler32_combine.argtypes = [u32, u32, u64]
    L.zgpu_host_alloc.restype = vp
    L.zgpu_host_alloc.argtypes = [u64]
    L.zgpu_host_free.restype = None
    L.zgpu_host_free.argtypes = [vp]
    L.zgpu_host_trim.restype = None
    L.zgpu_host_trim.argtypes = []
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise ZgpuError(rc, what)


def _u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def deflate_bound(n):
    return int(lib().zgpu_deflate_bound(n))


# the library's own search settings (zgpu_set_search documents them)
DEFAULT_MAX_CHAIN, DEFAULT_NICE_LENGTH = 8, 32


def set_search(max_chain=DEFAULT_MAX_CHAIN, nice_length=DEFAULT_NICE_LENGTH):
    """Engine-wide search settings; without arguments back to the library's defaults."""
    _check(lib().zgpu_set_search(max_chain, nice_length), "zgpu_set_search")


DEFAULT_DEEP_CHAIN, DEFAULT_PROBE_FANS = 96, 0


def set_long_match(deep_chain=DEFAULT_DEEP_CHAIN, probe_fans=DEFAULT_PROBE_FANS):
    """Bounds of the search kernel's long-match steps; without arguments back to the defaults."""
    _check(lib().zgpu_set_long_match(deep_chain, probe_fans), "zgpu_set_long_match")


class HostBuffer:
    """Page-locked host memory from zgpu_host_alloc as a numpy uint8 array (`.array`): input and output
    buffers of the host-buffer entry points that move at the full PCIe rate.  `close()` frees it; views
    of `.array` must not be used after that."""

    def __init__(self, nbytes):
        self.ptr = lib().zgpu_host_alloc(int(nbytes))
        if not self.ptr:
            raise ZgpuError(-3, "zgpu_host_alloc(%d)" % nbytes)  # ZGPU_E_NOMEM
        self.nbytes = int(nbytes)
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(1, self.nbytes)).from_address(self.ptr))[: self.nbytes]

    def close(self):
        if self.ptr:
            self.array = None
            lib().zgpu_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def host_trim():
    """Releases the spare staging buffers the devices keep between host-buffer calls."""
    lib().zgpu_host_trim()


def set_split(piece_bytes):
    _check(lib().zgpu_set_split(piece_bytes), "zgpu_set_split")


DEFAULT_DENSE_CHAIN = 2048


def set_dense_search(max_chain=DEFAULT_DENSE_CHAIN):
    """Chain budget of the second, reference-like search that units carried by a few byte values get
    (0 = no such pass); without arguments back to the default."""
    _check(lib().zgpu_set_dense_search(max_chain), "zgpu_set_dense_search")


def set_min_lazy(min_lazy):
    _check(lib().zgpu_set_min_lazy(min_lazy), "zgpu_set_min_lazy")


def partition_units(in_off, n_parts):
    """Unit ranges of a host batch over n_parts devices (host arithmetic, no GPU needed)."""
    off = np.ascontiguousarray(in_off, dtype=np.uint64)
    cut = np.zeros(n_parts + 1, dtype=np.uint32)
    _check(lib().zgpu_partition_units(off.ctypes.data, off.size - 1, n_parts, cut.ctypes.data), "zgpu_partition_units")
    return cut.tolist()


def crc32_combine(a, b, len_b):
    return int(lib().zgpu_crc32_combine(a, b, len_b))


def adler32_combine(a, b, len_b):
    return int(lib().zgpu_adler32_combine(a, b, len_b))


def checksum(data):
    """(crc32, adler32) of a host buffer, computed on the GPU."""
    a = _u8(data)
    c, ad = C.c_uint32(0), C.c_uint32(0)
    _check(lib().zgpu_checksum(a.ctypes.data if a.size else None, a.size, C.byref(c), C.byref(ad)),
           "zgpu_checksum")
    return c.value, ad.value


def crc32_update(data, crc):
    a = _u8(data)
    out = C.c_uint32(0)
    _check(lib().zgpu_crc32_update(a.ctypes.data if a.size else None, a.size, crc, C.byref(out)),
           "zgpu_crc32_update")
    return out.value


def adler32_update(data, adler):
    a = _u8(data)
    out = C.c_uint32(0)
    _check(lib().zgpu_adler32_update(a.ctypes.data if a.size else None, a.size, adler, C.byref(out)),
           "zgpu_adler32_update")
    return out.value


def checksum_device(ptr, nbytes, stream=None):
    c, ad = C.c_uint32(0), C.c_uint32(0)
    _check(lib().zgpu_checksum_device(ptr, nbytes, C.byref(c), C.byref(ad), stream), "zgpu_checksum_device")
    return c.value, ad.value


def checksum_kernel_ms():
    ms = C.c_float(0)
    _check(lib().zgpu_checksum_kernel_ms(C.byref(ms)), "zgpu_checksum_kernel_ms")
    return ms.value


def deflate_batch(units, ctype=DYNAMIC, lazy=0, flags=F_FINAL_EACH, want_crc=False, want_adler=False, history=None):
    """Compress a list of byte buffers in one GPU call (host buffers).

    history: the bytes of the same stream that precede the first unit (chunked streams with
    F_PRIME_WINDOW whose chunks are shared by several calls); up to 32 KiB of them are used.
    Returns (list of compressed bytes, crc32 list or None, adler32 list or None).
    """
    arrs = [_u8(u) for u in units]
    n = len(arrs)
    if n == 0:
        return [], ([] if want_crc else None), ([] if want_adler else None)
    hist = _u8(history)[-32768:] if history is not None and len(history) else None
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum([a.size for a in arrs], dtype=np.uint64)
    total = int(off[-1])
    if hist is not None:
        off += np.uint64(hist.size)
        flags |= F_PRIME_FIRST
        flat = np.concatenate([hist] + arrs)
    else:
        flat = np.concatenate(arrs) if total else np.zeros(1, dtype=np.uint8)
    out_off = np.zeros(n + 1, dtype=np.uint64)
    crc = np.zeros(n, dtype=np.uint32) if want_crc else None
    adl = np.zeros(n, dtype=np.uint32) if want_adler else None
    # two steps: the output is allocated once its size is known
    job, total = C.c_void_p(None), C.c_uint64(0)
    flags |= (F_WANT_CRC if want_crc else 0) | (F_WANT_ADLER if want_adler else 0)
    _check(lib().zgpu_deflate_batch_begin(flat.ctypes.data, off.ctypes.data, n, ctype, lazy, flags, C.byref(job),
                                          C.byref(total)), "zgpu_deflate_batch_begin")
    out = np.empty(max(1, total.value), dtype=np.uint8)
    rc = lib().zgpu_deflate_batch_finish(job, out.ctypes.data, out.size, out_off.ctypes.data,
                                         crc.ctypes.data if want_crc else None, adl.ctypes.data if want_adler else None)
    _check(rc, "zgpu_deflate_batch_finish")
    res = [out[int(out_off[i]):int(out_off[i + 1])].tobytes() for i in range(n)]
    return res, (crc.tolist() if want_crc else None), (adl.tolist() if want_adler else None)


class Plan:
    """Device-resident batch: unit table + workspace on the current HIP device."""

    def __init__(self, in_off, ctype=DYNAMIC, lazy=0, flags=F_FINAL_EACH):
        self.in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
        self.n_units = self.in_off.size - 1
        self._h = C.c_void_p(None)
        _check(lib().zgpu_plan_create(C.byref(self._h), self.in_off.ctypes.data, self.n_units, ctype, lazy,
                                      flags), "zgpu_plan_create")

    @property
    def out_bound(self):
        return int(lib().zgpu_plan_out_bound(self._h))

    def run(self, d_in, d_out, out_cap, stream=None):
        _check(lib().zgpu_plan_run(self._h, d_in, d_out, out_cap, stream), "zgpu_plan_run")

    def fetch(self, want_crc=False, want_adler=False, stream=None):
        off = np.zeros(self.n_units + 1, dtype=np.uint64)
        crc = np.zeros(self.n_units, dtype=np.uint32) if want_crc else None
        adl = np.zeros(self.n_units, dtype=np.uint32) if want_adler else None
        _check(lib().zgpu_plan_fetch(self._h, off.ctypes.data, crc.ctypes.data if want_crc else None,
                                     adl.ctypes.data if want_adler else None, stream), "zgpu_plan_fetch")
        return off, crc, adl

    def kernel_ms(self):
        a, b = C.c_float(0), C.c_float(0)
        _check(lib().zgpu_plan_kernel_ms(self._h, C.byref(a), C.byref(b)), "zgpu_plan_kernel_ms")
        return a.value, b.value

    def phase_ms(self):
        """(search kernel ms, emit kernel ms) of the last run's first round"""
        a, b = C.c_float(0), C.c_float(0)
        _check(lib().zgpu_plan_phase_ms(self._h, C.byref(a), C.byref(b)), "zgpu_plan_phase_ms")
        return a.value, b.value

    def out_off_device(self):
        return lib().zgpu_plan_out_off_device(self._h)

    def close(self):
        if self._h:
            lib().zgpu_plan_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
