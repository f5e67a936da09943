"""The strict decoder used as verifier: reference decode fixtures + round trips."""
import json
import os

import numpy as np
import pytest

import _oracle as zo

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dec():
    with open(os.path.join(HERE, "golden", "decode_fixtures.json")) as f:
        return json.load(f)


def test_reference_zlib_decode_fixtures(dec):
    # streams held by test/browser.es/inflate-test.js (pre-deflated data, issue#35)
    assert len(dec["zlib"]) == 4
    for f in dec["zlib"]:
        plain = zo.gen(f["plain"]).tobytes()
        assert zo.inflate_zlib(bytes.fromhex(f["stream_hex"]), len(plain)) == plain, f["name"]


def test_reference_gzip_decode_fixtures(dec):
    # streams held by test/browser.es/gunzip-test.js
    for f in dec["gzip"]:
        plain = zo.gen(f["plain"]).tobytes()
        assert zo.gunzip(bytes.fromhex(f["stream_hex"]), len(plain)) == plain, f["name"]


@pytest.mark.parametrize("ctype", [0, 1, 2])
def test_round_trip_named_cases(golden, ctype):
    for c in golden["cases"]:
        data = zo.gen(c["spec"])
        if data.size == 0 and ctype != 2:
            continue  # reference emits no block (NONE) / no EOB (FIXED) for empty input
        comp = zo.rawdeflate(data, ctype)
        got, used = zo.inflate_raw(comp, data.size, want_used=True)
        assert got == data.tobytes()
        assert used == len(comp)
        assert zo.inflate_zlib(zo.zlib_compress(data, ctype), data.size) == data.tobytes()


def test_round_trip_gzip():
    data = zo.gen({"gen": "text", "n": 20000, "seed": 1})
    g = zo.gzip_compress(data, 2, 0, mtime=123, flags=1 | 4, filename="x.txt")
    assert zo.gunzip(g, data.size) == data.tobytes()
    assert zo.gunzip(g + g, 2 * data.size) == data.tobytes() * 2


def test_strictness():
    data = zo.gen({"gen": "text", "n": 3000, "seed": 2}).tobytes()
    comp = bytearray(zo.rawdeflate(data, 0))
    comp[3] ^= 1  # NLEN mismatch, which the reference's reader does not notice (rawinflate.js:233)
    with pytest.raises(zo.InflateError) as e:
        zo.inflate_raw(bytes(comp), len(data))
    assert e.value.code == -3
    comp = zo.rawdeflate(data, 2)
    with pytest.raises(zo.InflateError):
        zo.inflate_raw(comp[:-5], len(data))
    with pytest.raises(zo.InflateError):
        zo.inflate_raw(b"\x07", 10)  # BTYPE 3
    z = bytearray(zo.zlib_compress(data, 2))
    z[-1] ^= 0xFF
    with pytest.raises(zo.InflateError) as e:
        zo.inflate_zlib(bytes(z), len(data))
    assert e.value.code == -9
    # empty FIXED from the reference has no end-of-block symbol: must be rejected
    with pytest.raises(zo.InflateError):
        zo.inflate_raw(zo.rawdeflate(b"", 1), 10)
