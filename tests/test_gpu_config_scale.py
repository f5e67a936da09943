"""Scaled-down runs of the larger benchmark configurations, decoded by Python's
own zlib / gzip / zipfile (decoders that share no code with this repository):
config 1 (1 MiB text, DYNAMIC: the reference's CPU-runnable case, size against
the reference), config 4 (zip archives of many 1-16 KiB files, alternating FIXED
and DYNAMIC per file as test/browser.es/zip-test.js does per entry) and config 5
(one gzip member made of independently compressed 1 MiB chunks, CRC-32 by
combine).  The unit counts are cut to what the CPU side verifies in seconds;
the layout of the work is the configurations' own.
"""
import gzip
import io
import zipfile
import zlib

import numpy as np
import pytest

import _oracle as zo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def z():
    import zlib_es_amd
    return zlib_es_amd


def _mixed(i, n):
    kind = i % 20
    if kind < 10:
        return zo.gen({"gen": "text", "n": n, "seed": 1000 + i})
    if kind < 15:
        return zo.gen({"gen": "rndseq", "n": n, "seed": 2000 + i})
    if kind < 18:
        return zo.gen({"gen": "rnd", "n": n, "seed": 3000 + i})
    return zo.gen({"gen": "runs", "n": n, "seed": 4000 + i})


def test_config1_text_1mib_against_reference(z):
    d = zo.gen({"gen": "text", "n": 1 << 20, "seed": 42})
    out = bytes(z.RawDeflate(d, {"compressionType": 2}).compress())
    assert zlib.decompress(out, wbits=-15) == d.tobytes()
    ref = len(zo.rawdeflate(d, 2))           # the reference's size, heap quirk included
    ref32 = len(zo.rawdeflate(d, 2, heap32=True))
    assert len(out) <= 1.03 * ref and len(out) <= 1.03 * ref32, (len(out), ref, ref32)


def test_config4_zip_many_small_files(z):
    rng = np.random.default_rng(4)
    n_arch, per = 3, 400
    total_in = total_out = 0
    for a in range(n_arch):
        zp = z.Zip()
        files = {}
        for f in range(per):
            n = int(rng.integers(1, 16385))
            data = _mixed(a * per + f, n)
            name = "a%d/f%04d.bin" % (a, f)
            files[name] = data.tobytes()
            zp.addFile(data, {"filename": list(name.encode()),
                              "deflateOption": {"compressionType": 1 if f & 1 else 2}})
        blob = bytes(zp.compress())
        total_in += sum(len(v) for v in files.values())
        total_out += len(blob)
        with zipfile.ZipFile(io.BytesIO(blob)) as zf:
            assert zf.testzip() is None  # CRC-32 of every entry
            assert sorted(zf.namelist()) == sorted(files)
            for name, want in files.items():
                assert zf.read(name) == want, name
    assert total_out < total_in  # the mix compresses


def test_config5_gzip_chunked_member(z):
    n_chunks, chunk = 24, 1 << 20
    parts = [_mixed(i, chunk) for i in range(n_chunks)]
    parts.append(zo.gen({"gen": "text", "n": 12345, "seed": 7}))  # ragged tail
    data = np.concatenate(parts)
    gz = bytes(z.gzip_chunked(data, chunk_size=chunk, mtime=0))
    assert gzip.decompress(gz) == data.tobytes()  # verifies CRC-32 and ISIZE
    # chunks that may not match into their predecessors: still one valid member, not smaller
    gz_ind = bytes(z.gzip_chunked(data, chunk_size=chunk, mtime=0, independent=True))
    assert gzip.decompress(gz_ind) == data.tobytes()
    assert len(gz) <= len(gz_ind)
    # header as the reference writes it (src/gzip.js:80-87): magic, CM=8, FLG=0, mtime, XFL=0, OS=255
    assert gz[:10] == bytes([0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 0, 255])
    # size against the reference on the same chunks (1 MiB units, summed)
    ref = sum(len(zo.rawdeflate(p, 2, heap32=True)) for p in parts[:4])
    ours, _, _ = z.deflate_batch(parts[:4], 2, 0)
    assert sum(len(o) for o in ours) <= 1.03 * ref


def test_maximum_size_unit(z):
    # one unit of ZGPU_MAX_UNIT_BYTES (256 MiB), resident on the device: cut into pieces by
    # default; every piece boundary, block boundary and ring wrap on the way is checked by
    # inflating the whole stream (on the CPU: one stream is serial work)
    import torch
    import bench
    N = z._native
    dev = torch.device("cuda:0")
    data = bench.gen_units(4096, 65536, 31, dev)
    n = data.numel()
    assert n == 256 << 20
    in_off = np.array([0, n], dtype=np.uint64)
    host = data.cpu().numpy()
    plan = z.Plan(in_off, 2, 0, N.F_FINAL_EACH | N.F_WANT_CRC)
    out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device=dev)
    plan.run(data.data_ptr(), out.data_ptr(), out.numel(), None)
    off, crc, _ = plan.fetch(want_crc=True)
    assert 0 < int(off[1]) <= z.deflate_bound(n)
    got, used = zo.inflate_raw(out[:int(off[1])].cpu().numpy(), n, want_used=True)
    assert used == int(off[1]) and got == host.tobytes()
    del got
    assert int(crc[0]) == N.checksum_device(data.data_ptr(), n)[0]
    del plan
    # a quarter of it as one piece (one workgroup walks 64 MiB: 1024 blocks, 1024 ring wraps)
    q = 64 << 20
    try:
        N.set_split(0)
        in_off = np.array([0, q], dtype=np.uint64)
        plan = z.Plan(in_off, 2, 0, N.F_FINAL_EACH)
        plan.run(data.data_ptr(), out.data_ptr(), out.numel(), None)
        off1, _, _ = plan.fetch()
        got, used = zo.inflate_raw(out[:int(off1[1])].cpu().numpy(), q, want_used=True)
        assert used == int(off1[1]) and got == host[:q].tobytes()
    finally:
        N.set_split(1 << 20)
