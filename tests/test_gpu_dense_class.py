"""Units carried by a few byte values, or made of short runs ("dense": every 4-byte key has
hundreds of earlier occurrences), are found on the device and searched the reference's way
(src/rawdeflate.js:370-395): their size is then within the bar of the reference's, which the
bounded default search misses by 3-47 %.  Everything else must not be touched by the pass."""
import numpy as np
import pytest

import _oracle as zo

pytestmark = pytest.mark.gpu

RATIO_BAR = 1.03


@pytest.fixture(scope="module")
def z():
    import torch
    assert torch.cuda.is_available()
    import zlib_es_amd as z
    return z


def _sizes(z, datas, ctype, lazy=0):
    outs, _, _ = z.deflate_batch(datas, ctype, lazy)
    for d, o in zip(datas, outs):
        assert zo.inflate_raw(o, d.size) == d.tobytes()
    return [len(o) for o in outs]


@pytest.mark.parametrize("ctype", [1, 2])
def test_small_alphabets_meet_the_bar(z, ctype):
    # 2..7 values, one unit each, with text and noise between them in the same batch
    datas, dense = [], []
    for k in range(2, 8):
        dense.append(len(datas))
        datas.append(zo.gen({"gen": "small-alpha", "n": 65536, "seed": 900 + k, "k": k}))
        datas.append(zo.gen({"gen": "text", "n": 30000 + k, "seed": k}))
        datas.append(zo.gen({"gen": "rnd", "n": 5000, "seed": k}))
    sizes = _sizes(z, datas, ctype)
    ratios = {len(np.unique(datas[i])): sizes[i] / len(zo.rawdeflate(datas[i], ctype)) for i in dense}
    assert max(ratios.values()) <= RATIO_BAR, ratios
    # ... which is the work of the second search: without it the same units are well outside
    z._native.set_dense_search(0)
    try:
        plain = _sizes(z, datas, ctype)
    finally:
        z._native.set_dense_search()
    assert plain[dense[0]] > 1.05 * sizes[dense[0]]
    assert all(plain[i] == sizes[i] for i in range(len(datas)) if i not in dense or len(np.unique(datas[i])) > 6)


def _families():
    rng = np.random.default_rng(2024)
    n = 65536

    def runs_of(k, maxlen):
        parts, total = [], 0
        while total < n:
            m = int(rng.integers(1, maxlen + 1))
            parts.append(np.full(m, (int(rng.integers(0, k)) * 37 + 1) & 255, dtype=np.uint8))
            total += m
        return np.concatenate(parts)[:n]

    def markov(k, stay):
        keep = rng.random(n) < stay
        nxt = rng.integers(0, k, n).astype(np.uint8)
        out = np.empty(n, dtype=np.uint8)
        v = 0
        for i in range(n):
            if not keep[i]:
                v = nxt[i]
            out[i] = v
        return out

    return {
        "two values 90:10": (rng.random(n) < 0.1).astype(np.uint8),
        "eight values, skewed": rng.choice(8, n, p=[0.6, 0.2, 0.08, 0.05, 0.03, 0.02, 0.01, 0.01]).astype(np.uint8),
        "markov 4 stay .8": markov(4, 0.8),
        "runs <= 8 of 3": runs_of(3, 8),
        "runs <= 16 of 8": runs_of(8, 16),
        "runs <= 32 of 4": runs_of(4, 32),
        # (the positions of a run are keyed on its value and length: up to 64 bytes of two values was 1.02-1.035 x
        # depending on the seed, up to 128 of four 1.05 x, up to 128 of two -- which are runs of up to several
        # hundred, as equal neighbours merge -- 1.08-1.12 x)
        "runs <= 64 of 2": runs_of(2, 64),
        "runs <= 128 of 4": runs_of(4, 128),
        "runs <= 128 of 2": runs_of(2, 128),
        "16-bit samples below 6": np.stack([rng.integers(0, 6, n // 2).astype(np.uint8), np.zeros(n // 2, dtype=np.uint8)], 1).reshape(-1),
    }


# (a skewed alphabet of many values -- geometric with p = 0.5: fifteen values in a sample -- is not taken: the class
# is bounded at eight distinct values so that zero-heavy binaries and indented text are not; 1.08 x under FIXED)
@pytest.mark.parametrize("ctype", [1, 2])
def test_skewed_alphabets_and_short_runs_meet_the_bar(z, ctype):
    fam = _families()
    datas = [np.ascontiguousarray(v) for v in fam.values()]
    sizes = _sizes(z, datas, ctype)
    ratios = {k: s / len(zo.rawdeflate(d, ctype)) for k, d, s in zip(fam, datas, sizes)}
    assert max(ratios.values()) <= RATIO_BAR, ratios


def test_lazy_is_the_callers_own_in_the_second_search(z):
    d = zo.gen({"gen": "small-alpha", "n": 40000, "seed": 31, "k": 3})
    for lazy in (0, 32):
        got = _sizes(z, [d], 2, lazy)[0]
        assert got <= RATIO_BAR * len(zo.rawdeflate(d, 2, lazy)), lazy


def test_other_content_is_not_touched(z):
    # the benchmark's kinds (among them long runs of four values), the named cases that are not small
    # alphabets, small-alphabet units below the classified size, and units of one value: same bytes with
    # the pass on and off
    import torch
    import bench
    rows = bench.gen_units(40, 65536, 7, torch.device("cuda")).view(-1, 65536).cpu().numpy()
    datas = [rows[i] for i in range(rows.shape[0])]
    datas += [zo.gen({"gen": "runs", "n": 65536, "seed": 3}), zo.gen({"gen": "rndseq", "n": 65536, "seed": 1}),
              zo.gen({"gen": "seq", "n": 76543}), zo.gen({"gen": "small-alpha", "n": 255, "seed": 1, "k": 2}),
              np.zeros(100000, dtype=np.uint8), np.full(65536, 0x41, dtype=np.uint8)]
    on, _, _ = z.deflate_batch(datas, 2, 0)
    z._native.set_dense_search(0)
    try:
        off, _, _ = z.deflate_batch(datas, 2, 0)
    finally:
        z._native.set_dense_search()
    assert [bytes(a) for a in on] == [bytes(b) for b in off]


def test_dense_units_in_every_round_and_piece(z, monkeypatch):
    # a token buffer that holds a few units at a time: every round classifies its own units; and a long
    # dense unit that is cut into pieces is classified piece by piece
    monkeypatch.setenv("ZGPU_TOK_BYTES", str(600000))
    datas = []
    for i in range(9):
        datas.append(zo.gen({"gen": "small-alpha", "n": 50000 + 1000 * i, "seed": 40 + i, "k": 2 + i % 3})
                     if i % 2 == 0 else zo.gen({"gen": "text", "n": 60000, "seed": i}))
    sizes = _sizes(z, datas, 1)
    for i in range(0, 9, 2):
        assert sizes[i] <= RATIO_BAR * len(zo.rawdeflate(datas[i], 1)), i
    monkeypatch.delenv("ZGPU_TOK_BYTES")
    big = zo.gen({"gen": "small-alpha", "n": 3 << 20, "seed": 77, "k": 4})
    z._native.set_dense_search(0)
    try:
        plain = _sizes(z, [big], 1)[0]
    finally:
        z._native.set_dense_search()
    assert _sizes(z, [big], 1)[0] < 0.95 * plain


def test_dense_small_files_among_others(z):
    # files of 1-16 KiB take the small-window variants of the first search by size class; the dense ones among
    # them still get the second search, which runs in the full-window variant
    rng = np.random.default_rng(8)
    datas, dense = [], []
    for i in range(300):
        n = int(rng.integers(1, 16385))
        if i % 7 == 3:
            datas.append(zo.gen({"gen": "small-alpha", "n": n, "seed": 500 + i, "k": 2 + i % 4}))
            if n >= 2000:  # (four and five values in 1000 bytes: 1.03-1.035 x)
                dense.append(i)
        else:
            datas.append(zo.gen({"gen": "text" if i % 2 else "rnd", "n": n, "seed": i}))
    for ctype in (1, 2):
        sizes = _sizes(z, datas, ctype)
        bad = {i: sizes[i] / len(zo.rawdeflate(datas[i], ctype)) for i in dense
               if sizes[i] > RATIO_BAR * len(zo.rawdeflate(datas[i], ctype))}
        assert not bad, (ctype, bad)


@pytest.mark.parametrize("ctype", [1, 2])
def test_device_resident_plan_on_a_callers_stream_gives_the_same_bytes(z, ctype, monkeypatch):
    # zgpu_plan_run searches the dense units beside the others (the first search runs on a stream of the plan's own);
    # the host-buffer call keeps both searches on one stream.  Same bytes either way, on a stream of the caller's,
    # run after run (the lists and queue heads start over), with other work queued on that stream before and after,
    # and with a token buffer that takes the batch in several rounds.
    import torch
    dev = torch.device("cuda")
    rng = np.random.default_rng(5)
    datas = []
    for i in range(48):
        kind = i % 4
        n = int(rng.integers(3000, 70000))
        if kind == 0:
            datas.append(zo.gen({"gen": "small-alpha", "n": n, "seed": 70 + i, "k": 2 + i % 7}))
        elif kind == 1:
            datas.append(zo.gen({"gen": "text", "n": n, "seed": i}))
        elif kind == 2:
            datas.append(np.repeat(rng.integers(0, 3, n // 4 + 1).astype(np.uint8) * 37, rng.integers(1, 9, n // 4 + 1))[:n])
        else:
            datas.append(zo.gen({"gen": "rnd", "n": n, "seed": i}))
    datas = [np.ascontiguousarray(d) for d in datas]
    want, _, _ = z.deflate_batch(datas, ctype, 0)
    off = np.zeros(len(datas) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([d.size for d in datas])
    flat = torch.from_numpy(np.concatenate(datas)).to(dev)
    for tok_bytes in (None, 700000):
        if tok_bytes:
            monkeypatch.setenv("ZGPU_TOK_BYTES", str(tok_bytes))
        plan = z.Plan(off, ctype, 0, 1)
        out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device=dev)
        stream = torch.cuda.Stream()
        noise = torch.empty(1 << 24, dtype=torch.float32, device=dev)
        for rep in range(3):
            with torch.cuda.stream(stream):
                noise.normal_()  # the caller's own work in front of the call ...
                plan.run(flat.data_ptr(), out.data_ptr(), out.numel(), stream.cuda_stream)
                noise.mul_(2.0)  # ... and behind it
                out_off, _, _ = plan.fetch(stream=stream.cuda_stream)
            host = out.cpu().numpy()
            for i, w in enumerate(want):
                assert bytes(host[int(out_off[i]):int(out_off[i + 1])]) == bytes(w), (tok_bytes, rep, i)
            out.zero_()
        plan.close()
        if tok_bytes:
            monkeypatch.delenv("ZGPU_TOK_BYTES")


def test_runs_in_long_units_and_primed_pieces(z):
    # the run keys of the thorough variant where positions wrap around the 64 KiB ring, where a piece's window
    # starts in the bytes before it, at the very start of a unit (nothing in front of the first run) and with runs
    # that reach the unit's end; the dense-class pass and the thorough search for everything
    rng = np.random.default_rng(77)

    def runs(n, k, maxlen, lead=0):
        lens = rng.integers(1, maxlen + 1, n // 2 + 2)
        vals = (rng.integers(0, k, lens.size) * 37 + 1).astype(np.uint8)
        d = np.repeat(vals, lens)[:n].copy()
        if lead:
            d[:lead] = d[0]
        return np.ascontiguousarray(d)

    datas = [runs(200000, 2, 64), runs(150001, 4, 128, lead=300), runs(70000, 3, 40), runs(65536 * 3 + 17, 2, 100),
             runs(9000, 2, 64), np.concatenate([runs(100000, 4, 64), np.full(5000, 38, dtype=np.uint8)]),
             runs(3 << 20, 4, 90)]
    for ctype in (1, 2):
        sizes = _sizes(z, datas, ctype)  # (round trip checked inside)
        for d, s in zip(datas[:6], sizes):
            assert s <= RATIO_BAR * len(zo.rawdeflate(d, ctype)), (ctype, d.size)
    z._native.set_search(32768, 258)
    try:
        for ctype in (1, 2):
            sizes = _sizes(z, datas[:6], ctype)
            for d, s in zip(datas[:6], sizes):
                assert s <= RATIO_BAR * len(zo.rawdeflate(d, ctype)), (ctype, d.size)
    finally:
        z._native.set_search()


def test_small_files_of_few_values_meet_the_bar(z):
    # files of 600-4000 bytes of two and three values, among text files of the same sizes: 1.07-1.20 x the reference's
    # size under FIXED while only units of 4 KiB and more were classified
    rng = np.random.default_rng(3)
    datas, dense = [], []
    for i in range(120):
        n = int(rng.integers(600, 4001))
        if i % 3 == 0:
            dense.append(i)
            datas.append(zo.gen({"gen": "small-alpha", "n": n, "seed": 900 + i, "k": 2 + (i // 3) % 2}))
        else:
            datas.append(zo.gen({"gen": "text", "n": n, "seed": i}))
    for ctype in (1, 2):
        sizes = _sizes(z, datas, ctype)
        got = sum(sizes[i] for i in dense)
        ref = sum(len(zo.rawdeflate(datas[i], ctype)) for i in dense)
        assert got <= RATIO_BAR * ref, (ctype, got / ref)
        worst = max(sizes[i] / len(zo.rawdeflate(datas[i], ctype)) for i in dense)
        assert worst <= 1.05, (ctype, worst)


def test_small_real_files_under_fixed(z):
    # sixteen C headers of the image cut to 1000 and to 3000 bytes: under FIXED every one was above 1.03 x the
    # reference's size at 1000 bytes (worst 1.045 x) and seven at 3000 while their chains were keyed on four bytes --
    # a small file lives on its three-byte matches, and the reference keys on three (src/rawdeflate.js:314-320)
    import glob
    import os
    files = [f for f in sorted(glob.glob("/usr/include/*.h")) if os.path.isfile(f) and os.path.getsize(f) >= 3000][:16]
    if len(files) < 8:
        pytest.skip("no C headers on this box")
    for n in (1000, 3000):
        datas = [np.fromfile(f, dtype=np.uint8, count=n) for f in files]
        for ctype in (1, 2):
            sizes = _sizes(z, datas, ctype)
            ratios = [s / len(zo.rawdeflate(d, ctype)) for s, d in zip(sizes, datas)]
            assert max(ratios) <= RATIO_BAR, (n, ctype, max(ratios))
