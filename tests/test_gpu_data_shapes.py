"""Shapes of real binary and source data that a search keyed by four bytes and bounded by its chain
table misses unless it looks for them (found in the image's own shared libraries and C headers,
unit by unit): arrays of 32-bit numbers that differ in their low byte, whose only matches are three
bytes long and lie one or two elements back (the reference keys on three bytes,
src/rawdeflate.js:370-395), and a preamble that repeats one file back, 29-32 KiB away (the
reference's window is the format's 32 KiB, src/rawdeflate.js:325-327)."""
import numpy as np
import pytest

import _oracle as zo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def z():
    import torch
    assert torch.cuda.is_available()
    import zlib_es_amd as z
    return z


def _ratio(z, data, ctype):
    outs, _, _ = z.deflate_batch([data], ctype, 0)
    assert zo.inflate_raw(outs[0], data.size) == data.tobytes()
    return len(outs[0]) / len(zo.rawdeflate(data, ctype))


def _table(seed, stride_words):
    # a symbol-table-like array: 32-bit little-endian numbers that mostly count up, some zero
    rng = np.random.default_rng(seed)
    n = 16384
    v = np.cumsum(rng.integers(1, 4, n)).astype(np.uint32) + np.uint32(0x2b00)
    v[rng.random(n) < 0.15] = 0
    if stride_words > 1:  # records of several words: the counting word, then words that repeat
        rec = np.zeros((n // stride_words, stride_words), dtype=np.uint32)
        rec[:, 0] = v[: n // stride_words]
        rec[:, 1:] = rng.integers(0, 1 << 31, stride_words - 1, dtype=np.uint32)[None, :]
        v = rec.reshape(-1)
    return np.frombuffer(v.astype("<u4").tobytes(), dtype=np.uint8).copy()


@pytest.mark.parametrize("ctype", [1, 2])
def test_arrays_of_small_numbers(z, ctype):
    # 1.3-1.4 x the reference's size without the three-byte matches at the records' strides
    for words in (1, 2, 3):
        r = _ratio(z, _table(5 + words, words), ctype)
        assert r <= 1.08, (words, r)


def _sources(seed, gap):
    # "files" of text, each starting with the same 700-byte preamble, `gap` bytes apart
    rng = np.random.default_rng(seed)
    pre = zo.gen({"gen": "text", "n": 700, "seed": 999}).tobytes()
    out = bytearray()
    k = 0
    while len(out) < 131072:
        body = zo.gen({"gen": "rnd", "n": gap - len(pre), "seed": seed * 100 + k}).tobytes()
        out += pre + bytes(b & 0x3f | 0x40 for b in body)  # 64 printable values: matches are rare in it
        k += 1
    return np.frombuffer(bytes(out[:131072]), dtype=np.uint8).copy()


@pytest.mark.parametrize("ctype", [1, 2])
def test_a_preamble_one_file_back(z, ctype):
    # the copies lie beyond the reach of the chain links (about 29,000 bytes) and inside the format's 32768
    for gap in (29500, 31000, 32700):
        r = _ratio(z, _sources(gap, gap), ctype)
        assert r <= 1.005, (gap, r)
