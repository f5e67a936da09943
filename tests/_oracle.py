"""ctypes binding of the CPU oracle (oracle/libzoracle.so).

Test infrastructure only: the oracle restates the reference's JavaScript
algorithms in C and holds the strict decoders used as verifier.  Imported by
tests, by ``__graft_entry__.smoke()`` and by ``bench.py``'s cpu_baseline leg,
never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

NONE, FIXED, DYNAMIC = 0, 1, 2


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORACLE_DIR, "libzoracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    u8p, sz, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
    L.zo_crc32_update.restype = u32
    L.zo_crc32_update.argtypes = [u8p, sz, u32]
    L.zo_crc32_single.restype = u32
    L.zo_crc32_single.argtypes = [u32, u32]
    L.zo_adler32_update.restype = u32
    L.zo_adler32_update.argtypes = [u32, u8p, sz]
    L.zo_lz77.restype = sz
    L.zo_lz77.argtypes = [u8p, sz, i32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.zo_get_lengths.restype = None
    L.zo_get_lengths.argtypes = [C.c_void_p, i32, i32, i32, C.c_void_p]
    L.zo_get_codes.restype = None
    L.zo_get_codes.argtypes = [C.c_void_p, i32, C.c_void_p]
    for name in ("zo_rawdeflate", "zo_rawdeflate_heap32"):
        f = getattr(L, name)
        f.restype = sz
        f.argtypes = [u8p, sz, i32, i32, u8p, sz, sz]
    L.zo_rawdeflate_many.restype = sz
    L.zo_rawdeflate_many.argtypes = [u8p, C.c_void_p, sz, i32, i32, u8p, sz]
    L.zo_deflate_bound.restype = sz
    L.zo_deflate_bound.argtypes = [sz]
    L.zo_zlib_compress.restype = sz
    L.zo_zlib_compress.argtypes = [u8p, sz, i32, i32, u8p, sz]
    L.zo_gzip_compress.restype = sz
    L.zo_gzip_compress.argtypes = [u8p, sz, i32, i32, u32, i32, C.c_char_p, C.c_char_p, u8p, sz]
    L.zo_inflate_raw.restype = i32
    L.zo_inflate_raw.argtypes = [u8p, sz, u8p, sz, C.POINTER(sz), C.POINTER(sz)]
    L.zo_inflate_raw_stats.restype = i32
    L.zo_inflate_raw_stats.argtypes = [u8p, sz, u8p, sz, C.POINTER(sz), C.POINTER(sz), C.c_void_p]
    L.zo_inflate_zlib.restype = i32
    L.zo_inflate_zlib.argtypes = [u8p, sz, u8p, sz, C.POINTER(sz)]
    L.zo_gunzip.restype = i32
    L.zo_gunzip.argtypes = [u8p, sz, u8p, sz, C.POINTER(sz)]
    for name in ("zo_gen_random", "zo_gen_random_sequential", "zo_gen_text", "zo_gen_runs"):
        f = getattr(L, name)
        f.restype = None
        f.argtypes = [u8p, sz, u32]
    L.zo_gen_sequential.restype = None
    L.zo_gen_sequential.argtypes = [u8p, sz]
    L.zo_gen_small_alpha.restype = None
    L.zo_gen_small_alpha.argtypes = [u8p, sz, u32, u32]
    _LIB = L
    return L


def _arr(data):
    """uint8 numpy array (contiguous) from bytes / bytearray / ndarray."""
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)


def _ptr(a):
    return a.ctypes.data if a.size else None


def crc32(data, crc=0):
    a = _arr(data)
    return lib().zo_crc32_update(_ptr(a), a.size, crc)


def adler32(data, adler=1):
    a = _arr(data)
    return lib().zo_adler32_update(adler, _ptr(a), a.size)


def rawdeflate(data, ctype=DYNAMIC, lazy=0, heap32=False, header=b""):
    a = _arr(data)
    cap = lib().zo_deflate_bound(a.size) + len(header)
    out = np.zeros(cap, dtype=np.uint8)
    out[: len(header)] = np.frombuffer(header, dtype=np.uint8)
    fn = lib().zo_rawdeflate_heap32 if heap32 else lib().zo_rawdeflate
    n = fn(_ptr(a), a.size, ctype, lazy, out.ctypes.data, cap, len(header))
    if n == C.c_size_t(-1).value:
        raise RuntimeError("oracle rawdeflate failed")
    return out[:n].tobytes()


def rawdeflate_many(flat, off, ctype=DYNAMIC, lazy=0):
    """Summed compressed size of the units flat[off[i]:off[i+1]] (one C call; releases the GIL)."""
    a = _arr(flat)
    o = np.ascontiguousarray(off, dtype=np.uint64)
    cap = lib().zo_deflate_bound(int((o[1:] - o[:-1]).max()) if o.size > 1 else 0) + 64
    scratch = np.empty(cap, dtype=np.uint8)
    r = lib().zo_rawdeflate_many(_ptr(a), o.ctypes.data, o.size - 1, ctype, lazy, scratch.ctypes.data, cap)
    if r == C.c_size_t(-1).value:
        raise RuntimeError("oracle rawdeflate_many failed")
    return r


def lz77(data, lazy=0):
    a = _arr(data)
    tok = np.zeros(2 * a.size + 8, dtype=np.uint16)
    fl = np.zeros(286, dtype=np.uint32)
    fd = np.zeros(30, dtype=np.uint32)
    n = lib().zo_lz77(_ptr(a), a.size, lazy, tok.ctypes.data, fl.ctypes.data, fd.ctypes.data)
    return tok[:n].copy(), fl, fd


def get_lengths(freqs, limit, heap16=True):
    f = np.ascontiguousarray(freqs, dtype=np.uint32)
    out = np.zeros(f.size, dtype=np.uint8)
    lib().zo_get_lengths(f.ctypes.data, f.size, limit, 1 if heap16 else 0, out.ctypes.data)
    return out


def get_codes(lengths):
    l = np.ascontiguousarray(lengths, dtype=np.uint8)
    out = np.zeros(l.size, dtype=np.uint16)
    lib().zo_get_codes(l.ctypes.data, l.size, out.ctypes.data)
    return out


def zlib_compress(data, ctype=DYNAMIC, lazy=0):
    a = _arr(data)
    cap = max(lib().zo_deflate_bound(a.size) + 16, 0x8000 + 16)
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().zo_zlib_compress(_ptr(a), a.size, ctype, lazy, out.ctypes.data, cap)
    if n == C.c_size_t(-1).value:
        raise RuntimeError("oracle zlib_compress failed")
    return out[:n].tobytes()


def gzip_compress(data, ctype=DYNAMIC, lazy=0, mtime=0, flags=0, filename=None, comment=None):
    a = _arr(data)
    cap = lib().zo_deflate_bound(a.size) + 64 + len(filename or "") + len(comment or "")
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().zo_gzip_compress(
        _ptr(a), a.size, ctype, lazy, mtime, flags,
        filename.encode("latin-1") if filename is not None else None,
        comment.encode("latin-1") if comment is not None else None,
        out.ctypes.data, cap)
    if n == C.c_size_t(-1).value:
        raise RuntimeError("oracle gzip_compress failed")
    return out[:n].tobytes()


class InflateError(Exception):
    def __init__(self, code):
        super().__init__("oracle inflate error %d" % code)
        self.code = code


def inflate_raw(data, cap, want_used=False, want_stats=False):
    a = _arr(data)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    ol, used = C.c_size_t(0), C.c_size_t(0)
    counts = np.zeros(3, dtype=np.uint32)
    e = lib().zo_inflate_raw_stats(_ptr(a), a.size, out.ctypes.data, cap, C.byref(ol), C.byref(used),
                                   counts.ctypes.data)
    if e:
        raise InflateError(e)
    res = out[: ol.value].tobytes()
    if want_stats:
        return res, used.value, counts
    return (res, used.value) if want_used else res


BLOCK_DTYPE = np.dtype([("litlen_len", np.uint8, 288), ("dist_len", np.uint8, 32),
                        ("litlen_freq", np.uint32, 288), ("dist_freq", np.uint32, 32)])


def block_stats(data, cap, max_blocks=64):
    """The dynamic blocks of a raw stream: per block the code lengths its header carries and the
    number of times each literal/length and distance symbol is used (a structured array)."""
    a = _arr(data)
    blocks = np.zeros(max_blocks, dtype=BLOCK_DTYPE)
    assert blocks.itemsize == 288 + 32 + 4 * 288 + 4 * 32
    L = lib()
    L.zo_inflate_record_blocks.argtypes = [C.c_void_p, C.c_int]
    L.zo_inflate_record_blocks.restype = None
    L.zo_inflate_record_blocks(blocks.ctypes.data, max_blocks)
    try:
        inflate_raw(a, cap)
        n = L.zo_inflate_recorded_blocks()
    finally:
        L.zo_inflate_record_blocks(None, 0)
    return blocks[:n]


def inflate_zlib(data, cap):
    a = _arr(data)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    ol = C.c_size_t(0)
    e = lib().zo_inflate_zlib(_ptr(a), a.size, out.ctypes.data, cap, C.byref(ol))
    if e:
        raise InflateError(e)
    return out[: ol.value].tobytes()


def gunzip(data, cap):
    a = _arr(data)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    ol = C.c_size_t(0)
    e = lib().zo_gunzip(_ptr(a), a.size, out.ctypes.data, cap, C.byref(ol))
    if e:
        raise InflateError(e)
    return out[: ol.value].tobytes()


def gen(spec):
    """Rebuild an input from a golden.json spec (twin of make() in gen_golden.mjs)."""
    g = spec["gen"]
    if g == "bytes":
        return np.frombuffer(bytes.fromhex(spec["hex"]), dtype=np.uint8).copy()
    n = spec["n"]
    out = np.zeros(n, dtype=np.uint8)
    p = out.ctypes.data if n else None
    seed = int(spec.get("seed", 0)) & 0xFFFFFFFF
    L = lib()
    if g == "rnd":
        L.zo_gen_random(p, n, seed)
    elif g == "seq":
        L.zo_gen_sequential(p, n)
    elif g == "rndseq":
        L.zo_gen_random_sequential(p, n, seed)
    elif g == "text":
        L.zo_gen_text(p, n, seed)
    elif g == "runs":
        L.zo_gen_runs(p, n, seed)
    elif g == "small-alpha":
        L.zo_gen_small_alpha(p, n, seed, spec["k"])
    else:
        raise ValueError(g)
    return out
