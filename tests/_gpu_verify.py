"""ctypes access to zlib.es_amd/libzgpu_verify.so, the GPU round-trip verifier
(include/zgpu_verify.h).  Test infrastructure: used by tests/ and bench.py only."""
import ctypes as C
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None

STATUS = ["ok", "truncated", "block type", "stored LEN/NLEN", "header", "code set", "symbol", "distance",
          "mismatch", "length", "trailing"]


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(ROOT, "zlib.es_amd", "libzgpu_verify.so"))
        L.zgpu_verify_streams.restype = C.c_int
        L.zgpu_verify_streams.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def verify(comp, comp_off, orig, orig_off, final_each=True, primed=False):
    """comp / orig: uint8 CUDA tensors; comp_off / orig_off: int64 or uint64 CUDA tensors of
    n+1 byte offsets.  Returns the per-unit status as a CPU int tensor (0 = ok)."""
    n = comp_off.numel() - 1
    status = torch.full((n,), 99, dtype=torch.int32, device=comp.device)
    rc = lib().zgpu_verify_streams(comp.data_ptr(), comp_off.data_ptr(), orig.data_ptr(), orig_off.data_ptr(), n,
                                   (1 if final_each else 0) | (2 if primed else 0), status.data_ptr(), None)
    if rc:
        raise RuntimeError("zgpu_verify_streams: hip error %d" % -rc)
    torch.cuda.synchronize()
    return status.cpu()
