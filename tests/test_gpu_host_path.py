"""The host-buffer entry points (what the N-API addon binds): sub-batches through the device's
stream slots, results against the device-resident plan path; streams continued over several
calls; the caller's lazy option."""
import ctypes as C

import numpy as np
import pytest

import _oracle as zo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def z():
    import torch
    torch.cuda.init()
    import zlib_es_amd
    return zlib_es_amd


def _mixed_units(n_units, seed):
    kinds = ("text", "rndseq", "rnd", "runs")
    rng = np.random.default_rng(seed)
    sizes = rng.integers(1, 200_000, n_units)
    return [zo.gen({"gen": kinds[i % 4], "n": int(s), "seed": seed + i}) for i, s in enumerate(sizes)]


def test_many_sub_batches_equal_one_plan(z):
    # 150 MiB in ragged units: five sub-batches over the four slots; the bytes must be what one
    # device-resident plan over the same units yields, whatever the cut into sub-batches
    import torch
    N = z._native
    units = _mixed_units(1500, 3)
    units.insert(700, np.zeros(0, dtype=np.uint8))
    units.insert(5, zo.gen({"gen": "text", "n": 3_000_000, "seed": 1}))  # cut into pieces in a small sub-batch only
    flat = np.concatenate(units)
    off = np.zeros(len(units) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([u.size for u in units])
    n = len(units)
    cap = sum(N.deflate_bound(u.size) + 64 for u in units)
    out = np.empty(cap, dtype=np.uint8)
    out_off = np.zeros(n + 1, dtype=np.uint64)
    crc = np.zeros(n, dtype=np.uint32)
    adler = np.zeros(n, dtype=np.uint32)
    for ctype in (2, 1, 0):
        rc = N.lib().zgpu_deflate_batch(flat.ctypes.data, off.ctypes.data, n, ctype, 0, N.F_FINAL_EACH, out.ctypes.data, cap,
                                        out_off.ctypes.data, crc.ctypes.data, adler.ctypes.data)
        assert rc == 0
        assert int(out_off[0]) == 0 and all(int(a) <= int(b) for a, b in zip(out_off, out_off[1:]))
        for i in (0, 1, 5, 6, 699, 700, 701, n - 1) + tuple(range(17, n, 97)):
            o = out[int(out_off[i]):int(out_off[i + 1])].tobytes()
            assert zo.inflate_raw(o, units[i].size) == units[i].tobytes(), (ctype, i)
            assert (int(crc[i]), int(adler[i])) == (zo.crc32(units[i]), zo.adler32(units[i])), (ctype, i)
        if ctype == 2:
            # the same units as one plan on the device (long units are cut by batch size there, so
            # compare the short ones byte for byte)
            plan = z.Plan(off, 2, 0, N.F_FINAL_EACH)
            d_in = torch.from_numpy(flat).cuda()
            d_out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device="cuda")
            plan.run(d_in.data_ptr(), d_out.data_ptr(), d_out.numel())
            poff, _, _ = plan.fetch()
            ref = d_out.cpu().numpy()
            for i in range(0, n, 13):
                if units[i].size < 250_000:
                    assert out[int(out_off[i]):int(out_off[i + 1])].tobytes() == \
                        ref[int(poff[i]):int(poff[i + 1])].tobytes(), i
    # two steps: the same bytes
    job, total = C.c_void_p(None), C.c_uint64(0)
    assert N.lib().zgpu_deflate_batch_begin(flat.ctypes.data, off.ctypes.data, n, 0, 0, N.F_FINAL_EACH, C.byref(job),
                                            C.byref(total)) == 0
    assert total.value == int(out_off[n])
    out2 = np.empty(total.value, dtype=np.uint8)
    off2 = np.zeros(n + 1, dtype=np.uint64)
    assert N.lib().zgpu_deflate_batch_finish(job, out2.ctypes.data, total.value, off2.ctypes.data, None, None) == 0
    assert off2.tolist() == out_off.tolist() and out2.tobytes() == out[:total.value].tobytes()


def test_stream_continued_over_calls_and_sub_batches(z):
    # chunks of one stream keep their window across sub-batch boundaries (inside one call) and
    # across calls (history): the concatenation is one valid stream, and a primed chunk is smaller
    N = z._native
    d = zo.gen({"gen": "text", "n": 80 * 1048576 // 8 + 4321, "seed": 5})
    chunk = 1 << 20
    chunks = [d[i:i + chunk] for i in range(0, d.size, chunk)]
    flags = N.F_SYNC_ALIGN | N.F_PRIME_WINDOW
    half = len(chunks) // 2
    a, ca, _ = z.deflate_batch(chunks[:half], 2, 0, flags, want_crc=True)
    cold, _, _ = z.deflate_batch(chunks[half:], 2, 0, flags | N.F_FINAL_LAST)
    b, cb, _ = z.deflate_batch(chunks[half:], 2, 0, flags | N.F_FINAL_LAST, want_crc=True,
                               history=np.concatenate(chunks[:half]))
    stream = b"".join(a) + b"".join(b)
    got, used = zo.inflate_raw(stream, d.size, want_used=True)
    assert got == d.tobytes() and used == len(stream)
    assert len(b[0]) < len(cold[0])  # the first chunk of the second call found matches in its history
    assert [int(c) for c in ca + cb] == [zo.crc32(c) for c in chunks]
    # the chunked zlib stream: header, body, Adler-32 by combine
    zs = z.zlib_chunked(d[:5_000_000], chunk_size=300_000)
    assert zo.inflate_zlib(zs, 5_000_000) == d[:5_000_000].tobytes()
    assert zs[:2] == zo.zlib_compress(d[:1000])[:2]
    assert zo.inflate_zlib(z.zlib_chunked(np.zeros(0, dtype=np.uint8)), 0) == b""


def test_lazy_option_is_the_deferral_threshold(z):
    # min_lazy 0: the caller's lazy alone decides (0 = the reference's greedy parse,
    # src/rawdeflate.js:339-347); sizes then differ with lazy and shrink as it grows
    N = z._native
    datas = [zo.gen({"gen": "text", "n": 65536, "seed": s}) for s in range(6)]
    try:
        N.set_min_lazy(0)
        sizes = {}
        for lazy in (0, 4, 16, 258):
            outs, _, _ = z.deflate_batch(datas, 2, lazy)
            for dd, o in zip(datas, outs):
                assert zo.inflate_raw(o, dd.size) == dd.tobytes()
            sizes[lazy] = sum(len(o) for o in outs)
        assert sizes[0] >= sizes[4] >= sizes[16] >= sizes[258] and sizes[0] > sizes[16] > sizes[258] * 0.9
        # the greedy parse against the reference's (same rule, bounded search): within the bar
        ref = sum(len(zo.rawdeflate(dd, 2, 0)) for dd in datas)
        assert sizes[0] <= 1.03 * ref
        N.set_min_lazy(258)
        outs, _, _ = z.deflate_batch(datas, 2, 0)
        assert sum(len(o) for o in outs) == sizes[258]  # the shipped default: always one step ahead
    finally:
        N.set_min_lazy(258)


def test_knobs_are_read_when_a_plan_is_created(z):
    import torch
    N = z._native
    d = zo.gen({"gen": "rndseq", "n": 65536, "seed": 3})
    off = np.array([0, d.size], dtype=np.uint64)
    d_in = torch.from_numpy(d).cuda()

    def run(plan):
        d_out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device="cuda")
        plan.run(d_in.data_ptr(), d_out.data_ptr(), d_out.numel())
        o, _, _ = plan.fetch()
        return int(o[1])

    try:
        N.set_long_match()  # the default walks on past the first long candidate
        deep_plan = z.Plan(off, 2, 0, N.F_FINAL_EACH)
        N.set_long_match(0, 1)
        fast_plan = z.Plan(off, 2, 0, N.F_FINAL_EACH)
        assert run(deep_plan) < run(fast_plan)  # the plan kept what it read, whatever was set later
    finally:
        N.set_long_match()


def test_batch_split_over_named_devices_on_the_one_gpu(z):
    """zgpu_init may name a device more than once: the batch is then split by bytes over two 'devices' that
    are the same GPU, so the partition, the per-device host threads and the concatenation of the multi-device
    path run for real on a one-GPU box.  The bytes must be those of the single-device call."""
    N = z._native
    L = N.lib()
    units = [zo.gen({"gen": ("text", "rndseq", "rnd", "runs")[i % 4], "n": 1 << 20, "seed": 40 + i}) for i in range(300)]  # 300 MiB
    flat = np.concatenate(units)
    n = len(units)
    off = (np.arange(n + 1, dtype=np.uint64) << np.uint64(20))
    cap = sum(N.deflate_bound(u.size) + 64 for u in units)

    def run(flags):
        out = np.empty(cap, dtype=np.uint8)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        crc = np.zeros(n, dtype=np.uint32)
        rc = L.zgpu_deflate_batch(flat.ctypes.data, off.ctypes.data, n, 2, 0, flags, out.ctypes.data, cap,
                                  out_off.ctypes.data, crc.ctypes.data, None)
        assert rc == 0
        return out[:int(out_off[n])].tobytes(), out_off.tolist(), crc.tolist()

    single = {f: run(f) for f in (N.F_FINAL_EACH, N.F_SYNC_ALIGN | N.F_FINAL_LAST | N.F_PRIME_WINDOW)}
    ids = (C.c_int32 * 2)(0, 0)
    assert L.zgpu_init(ids, 2) == 0
    try:
        got = (C.c_int32 * 8)()
        assert L.zgpu_host_devices(got, 8) == 2
        for f, ref in single.items():
            twice = run(f)
            assert twice[1] == ref[1] and twice[2] == ref[2], f
            assert twice[0] == ref[0], f
    finally:
        assert L.zgpu_init(None, 0) == 0
    # the chunked stream of the second flag set is one valid stream
    body = single[N.F_SYNC_ALIGN | N.F_FINAL_LAST | N.F_PRIME_WINDOW][0]
    assert zo.inflate_raw(body, flat.size) == flat.tobytes()


def test_spaced_finish_leaves_room_in_front_of_every_stream(z):
    N = z._native
    L = N.lib()
    units = _mixed_units(400, 9) + [np.zeros(0, dtype=np.uint8)]
    flat = np.concatenate(units)
    n = len(units)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum([u.size for u in units])
    job, total = C.c_void_p(None), C.c_uint64(0)
    assert L.zgpu_deflate_batch_begin(flat.ctypes.data, off.ctypes.data, n, 2, 0, N.F_FINAL_EACH | N.F_WANT_CRC,
                                      C.byref(job), C.byref(total)) == 0
    gaps = (np.arange(n, dtype=np.uint32) % 7) * 5 + 30
    room = int(total.value + gaps.sum()) + 100
    out = np.full(room, 0xA5, dtype=np.uint8)
    pos = np.zeros(n + 1, dtype=np.uint64)
    crc = np.zeros(n, dtype=np.uint32)
    assert L.zgpu_deflate_batch_finish_spaced(job, out.ctypes.data, room, gaps.ctypes.data, pos.ctypes.data,
                                              crc.ctypes.data, None) == 0
    assert int(pos[n]) == int(total.value + gaps.sum())
    assert np.all(out[int(pos[n]):] == 0xA5)
    prev_end = 0
    for i, u in enumerate(units):
        start = int(pos[i])
        assert start == prev_end + int(gaps[i])
        assert np.all(out[prev_end:start] == 0xA5), "a gap was written into"
        end = int(pos[i + 1]) - int(gaps[i + 1]) if i + 1 < n else int(pos[n])
        assert zo.inflate_raw(out[start:end].tobytes(), u.size) == u.tobytes(), i
        assert int(crc[i]) == zo.crc32(u)
        prev_end = end
