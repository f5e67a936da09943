"""Page-locked host memory for the host-buffer entry points (zgpu_host_alloc) and the staging the
library keeps between calls (zgpu_host_trim): same bytes whatever memory the caller's buffers are in."""
import ctypes

import numpy as np
import pytest

import _oracle as zo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def z():
    import zlib_es_amd
    return zlib_es_amd


def _batch(N, in_ptr, off, out_ptr, cap, flags):
    n = off.size - 1
    out_off = np.zeros(n + 1, dtype=np.uint64)
    crc = np.zeros(n, dtype=np.uint32)
    rc = N.lib().zgpu_deflate_batch(in_ptr, off.ctypes.data, n, 2, 0, flags, out_ptr, cap, out_off.ctypes.data,
                                    crc.ctypes.data, None)
    assert rc == 0, rc
    return out_off, crc


def test_page_locked_buffers_give_the_same_streams(z):
    N = z._native
    # 80 MiB in 20 units of ragged size: several sub-batches, so the gated upload / compute / download
    # stages of the host pipeline all run
    units = [zo.gen({"gen": ("text", "rndseq", "rnd", "runs")[i % 4], "n": (4 << 20) - 12345 * i, "seed": 90 + i})
             for i in range(20)]
    flat = np.concatenate(units)
    off = np.zeros(len(units) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([u.size for u in units])
    cap = int(sum(N.deflate_bound(u.size) + 48 for u in units)) + 64
    out_pg = np.zeros(cap, dtype=np.uint8)
    o1, c1 = _batch(N, flat.ctypes.data, off, out_pg.ctypes.data, cap, N.F_FINAL_EACH)
    pin_in, pin_out = N.HostBuffer(flat.size), N.HostBuffer(cap)
    try:
        pin_in.array[:] = flat
        o2, c2 = _batch(N, pin_in.ptr, off, pin_out.ptr, cap, N.F_FINAL_EACH)
        assert (o1 == o2).all() and (c1 == c2).all()
        assert bytes(pin_out.array[: int(o2[-1])]) == bytes(out_pg[: int(o1[-1])])
        o3, _ = _batch(N, pin_in.ptr, off, out_pg.ctypes.data, cap, N.F_FINAL_EACH)  # mixed: locked in, ordinary out
        assert (o3 == o1).all() and bytes(out_pg[: int(o3[-1])]) == bytes(pin_out.array[: int(o2[-1])])
    finally:
        pin_in.close()
        pin_out.close()
    for i, u in enumerate(units):
        assert zo.inflate_raw(out_pg[int(o1[i]):int(o1[i + 1])], u.size) == u.tobytes()
        assert int(c1[i]) == zo.crc32(u)


def test_trim_releases_the_spares_and_calls_go_on(z):
    N = z._native
    d = zo.gen({"gen": "text", "n": 3 << 20, "seed": 5})
    first = z.deflate_batch([d, d[:1000]], 2, 0)[0]
    N.host_trim()
    N.host_trim()  # nothing left to release: harmless
    again = z.deflate_batch([d, d[:1000]], 2, 0)[0]
    assert again == first


def test_init_with_a_bad_id_keeps_the_list_in_force(z):
    N = z._native
    L = N.lib()
    ids = (ctypes.c_int32 * 2)(0, 0)
    assert L.zgpu_init(ids, 2) == 0
    bad = (ctypes.c_int32 * 2)(0, 9999)
    assert L.zgpu_init(bad, 2) != 0
    got = (ctypes.c_int32 * 8)()
    assert L.zgpu_host_devices(got, 8) == 2 and list(got[:2]) == [0, 0]
    assert L.zgpu_init(None, 0) == 0
