"""The GPU round-trip verifier (include/zgpu_verify.h) against known-good and
deliberately damaged streams: it must accept what the CPU decoders accept and
name what is wrong with the rest.  Then every unit of a batch through it."""
import zlib

import numpy as np
import pytest
import torch

import _gpu_verify as gv
import _oracle as zo

pytestmark = pytest.mark.gpu


def _pack(chunks):
    off = np.zeros(len(chunks) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(c) for c in chunks])
    flat = np.frombuffer(b"".join(bytes(c) for c in chunks) or b"\0", dtype=np.uint8).copy()
    return torch.from_numpy(flat).cuda(), torch.from_numpy(off).cuda()


def _status(streams, origs, final_each=True):
    c, co = _pack(streams)
    o, oo = _pack(origs)
    return gv.verify(c, co, o, oo, final_each).tolist()


def test_accepts_reference_and_zlib_streams(golden):
    origs, streams = [], []
    for case in golden["cases"]:
        if case["n"] > 80000:
            continue
        d = zo.gen(case["spec"])
        for ct in (0, 1, 2):
            if d.size == 0 and ct != 2:
                continue  # the reference's output for empty input is not a valid stream there
            origs.append(d.tobytes())
            streams.append(zo.rawdeflate(d, ct))
        for level in (1, 6, 9):
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            origs.append(d.tobytes())
            streams.append(co.compress(d.tobytes()) + co.flush())
    assert _status(streams, origs) == [0] * len(streams)


def test_names_the_damage():
    d = zo.gen({"gen": "text", "n": 5000, "seed": 3}).tobytes()
    good = zlib.compressobj(6, zlib.DEFLATED, -15)
    good = good.compress(d) + good.flush()
    stored = b"\x01\x05\x00\xfa\xffhello"
    cases = [
        (good, d, 0),
        (good[:-3], d, gv.STATUS.index("truncated")),
        (good + b"\x00", d, gv.STATUS.index("trailing")),
        (good, d[:-1] + b"!", None),                      # mismatch or length, never ok
        (good, d + b"x", gv.STATUS.index("length")),
        (stored, b"hello", 0),
        (b"\x01\x05\x00\xfa\xfehello", b"hello", gv.STATUS.index("stored LEN/NLEN")),
        (b"\x07", b"", gv.STATUS.index("block type")),
        (stored, b"hellp", gv.STATUS.index("mismatch")),
    ]
    got = _status([c[0] for c in cases], [c[1] for c in cases])
    for (s, o, want), g in zip(cases, got):
        if want is None:
            assert g != 0
        else:
            assert g == want, (gv.STATUS[g], gv.STATUS[want])
    # flipping any single bit of a short dynamic block never yields "ok" with the same bytes
    short = zlib.compressobj(9, zlib.DEFLATED, -15)
    src = b"abracadabra, abracadabra! " * 8
    short = short.compress(src) + short.flush()
    flips = []
    for i in range(len(short) * 8):
        b = bytearray(short)
        b[i >> 3] ^= 1 << (i & 7)
        flips.append(bytes(b))
    st = _status(flips, [src] * len(flips))
    # a flip in the final byte's padding bits leaves the stream unchanged in meaning
    assert sum(1 for x in st if x == 0) <= 7


def test_every_unit_of_a_batch():
    import zlib_es_amd as z
    datas = [zo.gen({"gen": g, "n": n, "seed": i}) for i, (g, n) in enumerate(
        [("text", 65536), ("rndseq", 65536), ("rnd", 65536), ("runs", 65536), ("text", 0), ("text", 1),
         ("text", 300000), ("rnd", 70001)] * 8)]
    for ct in (1, 2):
        outs, _, _ = z.deflate_batch(datas, ct, 0)
        assert _status(outs, [d.tobytes() for d in datas]) == [0] * len(datas)
    # chunks of one stream: all but the last end with an empty stored block
    N = z._native
    outs, _, _ = z.deflate_batch(datas[:8], 2, 0, flags=N.F_SYNC_ALIGN | N.F_FINAL_LAST)
    assert _status(outs, [d.tobytes() for d in datas[:8]], final_each=False) == [0] * 8
    assert _status(outs, [d.tobytes() for d in datas[:8]], final_each=True)[0] != 0
