"""Randomised batches through the whole GPU path, every unit checked on the
device by the round-trip verifier and every requested checksum against the
oracle: unit sizes from 0 to a few blocks, all content kinds, both Huffman
modes, lazy settings, independent streams and chunked streams, several search
depths.  The encoder's waves hand work to each other through LDS flags; a
batch of a few thousand units of uneven sizes is what shakes that out."""
import os

import numpy as np
import pytest
import torch

import _gpu_verify as gv
import _oracle as zo

pytestmark = pytest.mark.gpu

KINDS = ["text", "rndseq", "rnd", "runs", "seq", "small-alpha"]


def _unit(rng, i):
    kind = KINDS[int(rng.integers(0, len(KINDS)))]
    r = rng.random()
    if r < 0.15:
        n = int(rng.integers(0, 70))
    elif r < 0.6:
        n = int(rng.integers(70, 20000))
    elif r < 0.95:
        n = int(rng.integers(20000, 140000))
    else:
        n = int(rng.integers(140000, 400000))
    spec = {"gen": kind, "n": n, "seed": int(rng.integers(1, 1 << 30))}
    if kind == "small-alpha":
        spec["k"] = int(rng.integers(1, 5))
    return zo.gen(spec)


# ZGPU_FUZZ_SEEDS="a-b" widens the run (default: three seeds, about ten seconds)
_lo, _hi = (int(x) for x in os.environ.get("ZGPU_FUZZ_SEEDS", "1-3").split("-"))


@pytest.mark.parametrize("seed", list(range(_lo, _hi + 1)))
def test_random_batches(seed):
    import zlib_es_amd as z
    N = z._native
    rng = np.random.default_rng(seed)
    datas = [_unit(rng, i) for i in range(600)]
    origs = [d.tobytes() for d in datas]
    flat = np.concatenate(datas)
    off = np.zeros(len(datas) + 1, dtype=np.int64)
    off[1:] = np.cumsum([d.size for d in datas])
    d_orig = torch.from_numpy(flat).cuda()
    d_ooff = torch.from_numpy(off).cuda()
    crc_ref = [zo.crc32(d) for d in datas[:50]]
    adl_ref = [zo.adler32(d) for d in datas[:50]]
    try:
        # chunked: 1 = units are chunks of one stream, 2 = and each may match into the 32 KiB before it
        # the last two settings add the long-match steps (deep walk 128 / 1024, two probe-ahead fans):
        # units of every size class wait for the links of the items ahead of them there
        for ctype, lazy, chain, nice, chunked, deep, fan in [
                (2, 0, 8, 32, 0, 0, 1), (1, 0, 8, 258, 0, 0, 1), (2, 16, 3, 32, 0, 0, 1), (2, 0, 32, 258, 1, 0, 1),
                (1, 7, 1, 8, 1, 0, 1), (2, 0, 8, 32, 2, 0, 1), (1, 3, 5, 40, 2, 0, 1), (2, 0, 8, 32, 0, 128, 2),
                (1, 0, 4, 32, 2, 1024, 2)]:
            z.set_search(chain, nice)
            N.set_long_match(deep, fan)
            flags = (N.F_SYNC_ALIGN | N.F_FINAL_LAST) if chunked else N.F_FINAL_EACH
            if chunked == 2:
                flags |= N.F_PRIME_WINDOW
            outs, crc, adl = z.deflate_batch(datas, ctype, lazy, flags=flags, want_crc=True, want_adler=True)
            assert crc[:50] == crc_ref and adl[:50] == adl_ref
            co = np.zeros(len(outs) + 1, dtype=np.int64)
            co[1:] = np.cumsum([len(o) for o in outs])
            d_comp = torch.from_numpy(np.frombuffer(b"".join(outs), dtype=np.uint8).copy()).cuda()
            st = gv.verify(d_comp, torch.from_numpy(co).cuda(), d_orig, d_ooff, final_each=not chunked,
                           primed=chunked == 2)
            bad = [(i, gv.STATUS[int(s)], datas[i].size) for i, s in enumerate(st.tolist()) if s]
            assert not bad, (ctype, lazy, chain, nice, chunked, deep, fan, bad[:5])
            if chunked:
                stream = b"".join(outs)
                assert zo.inflate_raw(stream, flat.size) == flat.tobytes()
            assert all(len(o) <= z.deflate_bound(len(d)) for o, d in zip(outs, origs))
    finally:
        z.set_search()
        N.set_long_match()


def _skewed_unit(rng):
    """A unit whose symbol statistics are geometric: literal values, match lengths and match
    distances each follow counts that grow by a random factor, so the three Huffman trees of a
    block (and the code-length tree of its header) come out anywhere from flat to chains far
    deeper than their length limits."""
    n = int(rng.integers(2000, 66000))
    nsym = int(rng.integers(2, 257))
    r = float(rng.uniform(1.05, 2.3))
    w = np.minimum(r ** np.arange(nsym), 1e6)
    w = np.maximum(1, (w * (n / w.sum())).astype(np.int64))
    vals = rng.permutation(256)[:nsym].astype(np.uint8)
    d = np.repeat(vals, w)
    rng.shuffle(d)
    d = d[:n].copy()
    n = d.size
    # planted copies: lengths and distances from geometric class counts
    lens = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
    dists = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
             4097, 6145, 8193, 12289, 16385, 24577]
    rl, rd = float(rng.uniform(1.0, 2.2)), float(rng.uniform(1.0, 2.2))
    pl = rl ** rng.permutation(len(lens))
    pd = rd ** rng.permutation(len(dists))
    pl, pd = pl / pl.sum(), pd / pd.sum()
    cur = 1
    budget = int(rng.integers(0, 400))
    while budget > 0 and cur + 300 < n:
        ln = lens[int(rng.choice(len(lens), p=pl))]
        di = dists[int(rng.choice(len(dists), p=pd))]
        cur += int(rng.integers(1, 40))
        if di <= cur and cur + ln < n:
            for k in range(ln):  # byte by byte: a copy may overlap itself
                d[cur + k] = d[cur + k - di]
            cur += ln
        budget -= 1
    return d


_slo, _shi = (int(x) for x in os.environ.get("ZGPU_FUZZ_SKEW_SEEDS", "11-12").split("-"))


@pytest.mark.parametrize("seed", list(range(_slo, _shi + 1)))
def test_skewed_symbol_statistics(seed):
    import zlib_es_amd as z
    rng = np.random.default_rng(seed)
    datas = [_skewed_unit(rng) for _ in range(400)]
    flat = np.concatenate(datas)
    off = np.zeros(len(datas) + 1, dtype=np.int64)
    off[1:] = np.cumsum([d.size for d in datas])
    d_orig = torch.from_numpy(flat).cuda()
    d_ooff = torch.from_numpy(off).cuda()
    try:
        for chain, nice, lazy in ((8, 32, 0), (64, 258, 0), (4, 32, 8)):
            z.set_search(chain, nice)
            outs, _, _ = z.deflate_batch(datas, 2, lazy)
            co = np.zeros(len(outs) + 1, dtype=np.int64)
            co[1:] = np.cumsum([len(o) for o in outs])
            d_comp = torch.from_numpy(np.frombuffer(b"".join(outs), dtype=np.uint8).copy()).cuda()
            st = gv.verify(d_comp, torch.from_numpy(co).cuda(), d_orig, d_ooff)
            bad = [(i, gv.STATUS[int(s)], datas[i].size) for i, s in enumerate(st.tolist()) if s]
            assert not bad, (chain, nice, lazy, bad[:5])
            for i in range(0, len(outs), 40):  # and a sample through the CPU decoder
                assert zo.inflate_raw(outs[i], datas[i].size) == datas[i].tobytes()
    finally:
        z.set_search()
