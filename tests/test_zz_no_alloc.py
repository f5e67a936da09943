"""The checksum entry point that allocates nothing (zgpu_checksum_device_noalloc): called through
the C ABI, checked against zgpu_checksum_device and the CPU oracle, twice in a row with the second
input smaller, so that whatever the first call left in the library's fixed device array must not
show in the second result."""
import ctypes as C

import numpy as np
import pytest

import _oracle as zo

pytestmark = pytest.mark.gpu


def test_noalloc_checksum_matches_the_allocating_entry_point():
    import torch
    torch.cuda.init()
    import zlib_es_amd as z
    N = z._native
    L = N.lib()
    rng = np.random.default_rng(5)
    big = rng.integers(0, 256, (37 << 20) + 12345, dtype=np.uint8)
    big[1 << 20: 3 << 20] = 0xFF
    small = big[7: 7 + (2 << 20) + 3].copy()
    tiny = big[:61].copy()
    out = torch.zeros(2, dtype=torch.int32, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.zgpu_noalloc_setup() == 0
    # every call below must leave the device's memory in use where it is: no allocation, no release
    for data in (big, small, tiny, np.zeros(0, dtype=np.uint8), small):
        t = torch.from_numpy(data).cuda() if data.size else torch.zeros(1, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        rc = L.zgpu_checksum_device_noalloc(C.c_void_p(t.data_ptr()), data.size, C.c_void_p(out.data_ptr()), stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before
        got = tuple(int(v) & 0xFFFFFFFF for v in out.cpu().tolist())
        assert got == N.checksum_device(t.data_ptr(), data.size, stream), data.size
        assert got == (zo.crc32(data), zo.adler32(data)), data.size
    # beyond the fixed array's reach: a clear error, nothing launched
    assert L.zgpu_checksum_device_noalloc(C.c_void_p(out.data_ptr()), L.zgpu_noalloc_max_bytes() + 1,
                                          C.c_void_p(out.data_ptr()), stream) == -6
    assert L.zgpu_checksum_device_noalloc(None, 16, C.c_void_p(out.data_ptr()), stream) == -2
