"""The N > 1 path on CPU: two gloo ranks shard units, compress their ranges
(the GPU call is replaced by the oracle through `compress_fn`, the documented
test hook) and rank 0 reassembles; partitioning and checksum combine are pure
host logic."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import _oracle as zo


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


HISTORIES = []  # what the sharder passed as history, per call (this process)


def _oracle_fn(units, ctype, lazy, flags, history=None):
    # stands in for the GPU: complete streams per unit (FINAL_EACH) or stored-aligned chunks
    HISTORIES.append(None if history is None else bytes(np.asarray(history, dtype=np.uint8)))
    outs, crcs, adlers = [], [], []
    for i, u in enumerate(units):
        u = np.asarray(u, dtype=np.uint8)
        raw = zo.rawdeflate(u, ctype, lazy)
        final = (flags & 1) or ((flags & 4) and i == len(units) - 1) or not (flags & 2)
        if not final:
            # a chunk that is followed by more of its stream: stored blocks are byte-aligned, so
            # clearing BFINAL of the (single) block is all it takes
            assert ctype == 0 and u.size <= 65535
            b = bytearray(raw)
            b[0] &= 0xFE
            raw = bytes(b)
        outs.append(raw)
        crcs.append(zo.crc32(u))
        adlers.append(zo.adler32(u))
    return outs, crcs, adlers


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import zlib_es_amd as z
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        units = [zo.gen({"gen": "text", "n": 1000 + 777 * i, "seed": i}) for i in range(11)]
        res = z.compress_sharded(units, 2, 0, compress_fn=_oracle_fn)
        if rank == 0:
            outs, crcs, adlers = res
            ok = all(zo.inflate_raw(o, u.size) == u.tobytes() for o, u in zip(outs, units))
            ok = ok and crcs == [zo.crc32(u) for u in units] and adlers == [zo.adler32(u) for u in units]
            q.put(("units", ok, len(outs)))
        else:
            assert res is None
        # chunks of one stream: gzip and zlib framing around the sharded body, and the bytes before a
        # rank's first chunk handed over as history so that its window is primed across the boundary
        d = zo.gen({"gen": "text", "n": 300_000, "seed": 77})
        del HISTORIES[:]
        gz = z.gzip_chunked(d, chunk_size=50_000, ctype=0, compress_fn=_oracle_fn)
        zs = z.zlib_chunked(d, chunk_size=50_000, ctype=0, compress_fn=_oracle_fn)
        if rank == 0:
            assert HISTORIES == [None, None]
            q.put(("streams", zo.gunzip(gz, d.size) == d.tobytes() and zo.inflate_zlib(zs, d.size) == d.tobytes()
                   and zs[:2] == zo.zlib_compress(d, 0)[:2], 0))
        else:
            assert gz is None and zs is None
            lo = z.partition_by_bytes([50_000] * 6, 2)[1][0]
            assert HISTORIES == [d[lo * 50_000 - 32768:lo * 50_000].tobytes()] * 2
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_partition_by_bytes():
    import zlib_es_amd as z
    sizes = [10, 0, 5, 100, 1, 1, 1, 50, 3]
    for world in (1, 2, 3, 4, 8, 16):
        parts = z.partition_by_bytes(sizes, world)
        assert len(parts) == world
        assert parts[0][0] == 0 and parts[-1][1] == len(sizes)
        for (a, b), (c, d) in zip(parts, parts[1:]):
            assert b == c and a <= b
    parts = z.partition_by_bytes([64] * 1024, 8)
    assert all(hi - lo == 128 for lo, hi in parts)
    assert z.partition_by_bytes([], 4) == [(0, 0)] * 4


def test_combine_checksums_fold():
    import zlib_es_amd as z
    d = zo.gen({"gen": "rnd", "n": 100_000, "seed": 4})
    cuts = [0, 1, 4096, 4097, 70_000, 100_000]
    chunks = [d[a:b] for a, b in zip(cuts, cuts[1:])]
    crc, ad = z.combine_checksums([zo.crc32(c) for c in chunks], [zo.adler32(c) for c in chunks],
                                  [c.size for c in chunks])
    assert (crc, ad) == (zo.crc32(d), zo.adler32(d))


def test_two_rank_sharding_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    kind, ok, n = q.get(timeout=10)
    assert kind == "units" and ok and n == 11
    kind, ok, _ = q.get(timeout=10)
    assert kind == "streams" and ok


def test_c_partition_for_1_2_4_8_devices():
    # the split zgpu_deflate_batch makes over the devices named by zgpu_init (host arithmetic)
    import zlib_es_amd as z
    rng = np.random.default_rng(1)
    for sizes in ([65536] * 1024, rng.integers(1, 16385, 5000).tolist(), [0, 0, 7, 0, 100, 0], [5], [10] * 3):
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64) + np.uint64(12345)
        total = int(off[-1] - off[0])
        for parts in (1, 2, 4, 8):
            cut = z._native.partition_units(off, parts)
            assert cut[0] == 0 and cut[-1] == len(sizes) and all(a <= b for a, b in zip(cut, cut[1:]))
            if len(sizes) >= 1024:
                per = [int(off[b] - off[a]) for a, b in zip(cut, cut[1:])]
                assert max(per) - min(per) <= 2 * max(sizes), (parts, per)
                assert sum(per) == total
