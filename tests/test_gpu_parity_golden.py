"""The stream fixture is current: the streams the product path produces now, for the fixture's
named inputs and settings, equal tests/golden/gpu_streams.json byte for byte.  The fixture is
what the reference's own decoders are run on (where the reference is, no GPU is); this test
ties it to the kernels as they are.  On a mismatch: run tools/make_gpu_streams.py on the GPU,
copy gpurun_out/gpu_streams.json to tests/golden/, and run tests/golden/check_gpu_streams.mjs
where the reference tree is."""
import json
import os

import pytest

import _gpu_streams

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gpu_streams.json")


def _key(s):
    return json.dumps({k: v for k, v in s.items() if k not in ("b64",)}, sort_keys=True)


@pytest.mark.gpu
def test_regenerated_streams_equal_the_fixture():
    want = json.load(open(FIX))["streams"]
    # the Gzip header holds the time of the call unless mtime is given: compare those past the header
    got = _gpu_streams.make_streams()["streams"]
    assert [_key(s) for s in got] == [_key(s) for s in want], "the fixture's list of streams differs: regenerate it"
    import base64
    bad = []
    for g, w in zip(got, want):
        a, b = base64.b64decode(g["b64"]), base64.b64decode(w["b64"])
        if g["kind"] == "gzip" and "chunked" not in g:
            # MTIME (bytes 4..8) is the wall clock; with FHCRC the header CRC-16 follows it
            a, b = _strip_gzip_header(a), _strip_gzip_header(b)
        if a != b:
            bad.append(_key(g))
    assert not bad, "streams differ from the fixture (regenerate it and re-run the reference decoders): %s" % bad[:5]


def _strip_gzip_header(m):
    flg = m[3]
    p = 10
    if flg & 4:
        p += 2 + (m[p] | m[p + 1] << 8)
    if flg & 8:
        p = m.index(0, p) + 1
    if flg & 16:
        p = m.index(0, p) + 1
    if flg & 2:
        p += 2
    return m[:4] + m[8:p - (2 if flg & 2 else 0)] + m[p:]
