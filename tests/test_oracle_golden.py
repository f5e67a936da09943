"""Pins the CPU oracle against outputs of the reference itself.

tests/golden/golden.json was produced by tests/golden/gen_golden.mjs, which
runs the reference's per-module sources (src/rawdeflate.js, src/crc32.js,
src/adler32.js, src/deflate.js, src/gzip.js) under Node.  Every comparison
here is byte-exact (sha256 / hex / integer equality).
"""
import hashlib

import numpy as np
import pytest

import _oracle as zo


def _sha(b):
    return hashlib.sha256(b).hexdigest()


def _check(rec, got):
    assert len(got) == rec["len"]
    assert _sha(got) == rec["sha256"]
    if "hex" in rec:
        assert got.hex() == rec["hex"]


def test_checksums_named_cases(golden):
    for c in golden["cases"]:
        data = zo.gen(c["spec"])
        assert data.size == c["n"]
        assert zo.crc32(data) == c["crc32"], c["spec"]
        assert zo.adler32(data) == c["adler32"], c["spec"]


def test_checksum_known_answers():
    # values asserted by the reference's own tests / survey KATs
    assert zo.crc32(b"123456789") == 0xCBF43926
    assert zo.crc32(b"hoge\n") == 0xB9E2879D  # test/browser.es/unzip-test.js stored entry
    assert zo.adler32(b"Wikipedia") == 0x11E60398
    assert zo.adler32(bytes(range(1, 24))) == 0x09130115


def test_checksum_resumable():
    rng = np.random.default_rng(1)
    data = rng.integers(0, 256, 100_000, dtype=np.uint8)
    for cut in (0, 1, 7, 1023, 1024, 1025, 65521, 99_999):
        assert zo.crc32(data[cut:], zo.crc32(data[:cut])) == zo.crc32(data)
        assert zo.adler32(data[cut:], zo.adler32(data[:cut])) == zo.adler32(data)


@pytest.mark.parametrize("ctype", [0, 1, 2])
def test_rawdeflate_named_cases(golden, ctype):
    for c in golden["cases"]:
        data = zo.gen(c["spec"])
        got = zo.rawdeflate(data, ctype)
        rec = c["raw"][str(ctype)]
        if c["n"] == 0 and ctype == 0:
            # reference returns its untouched 32 KiB zero buffer (rawdeflate.js:106,123)
            assert rec["len"] == 0x8000 and got == b""
            continue
        _check(rec, got)


def test_lz77_tokens(golden):
    n = 0
    for c in golden["cases"]:
        if "tokens" not in c:
            continue
        tok, _, _ = zo.lz77(zo.gen(c["spec"]))
        assert tok.tolist() == c["tokens"], c["spec"]
        n += 1
    assert n >= 8
    tok, _, _ = zo.lz77(b"abcabcabcabcXabcabc")
    assert tok.tolist() == [97, 98, 99, 263, 0, 0, 2, 0, 0, 88, 260, 0, 0, 5, 0, 1, 256]


def test_lazy_variants(golden):
    for r in golden["lazy"]:
        got = zo.rawdeflate(zo.gen(r["spec"]), r["ctype"], r["lazy"])
        _check(r["out"], got)


def test_fuzz(golden):
    for r in golden["fuzz"]:
        data = zo.gen(r["spec"])
        assert zo.crc32(data) == r["crc32"]
        assert zo.adler32(data) == r["adler32"]
        for ct in (0, 1, 2):
            got = zo.rawdeflate(data, ct, r["lazy"])
            if data.size == 0 and ct == 0:
                continue
            assert len(got) == r["l%d" % ct], (r["spec"], ct)
            assert _sha(got)[:16] == r["s%d" % ct], (r["spec"], ct)


def test_code_lengths_and_codes(golden):
    for t in golden["lengths"]:
        l = zo.get_lengths(t["freqs"], t["limit"])
        assert l.tolist() == t["lengths"]
        assert zo.get_codes(l).tolist() == t["codes"]


def test_zlib_container(golden):
    for c in golden["cases"]:
        data = zo.gen(c["spec"])
        for ct in (0, 1, 2):
            _check(c["zlib"][str(ct)], zo.zlib_compress(data, ct))


def test_gzip_container(golden):
    for g in golden["gzip"]:
        opt = g["opt"]
        fl = opt.get("flags", {})
        flags = (1 if fl.get("fname") else 0) | (2 if fl.get("fcomment") else 0) | \
                (4 if fl.get("fhcrc") else 0) | (8 if fl.get("comment") else 0)
        do = opt.get("deflateOptions", {})
        got = zo.gzip_compress(zo.gen(g["spec"]), do.get("compressionType", 2), do.get("lazy", 0),
                               g["mtime"], flags, opt.get("filename"), opt.get("comment"))
        _check(g["out"], got)


def test_heap16_quirk_is_reachable():
    # a frequency of exactly 65536 is stored as 0 in the Uint16 heap
    f = np.zeros(286, dtype=np.uint32)
    f[10], f[11], f[256] = 65536, 3, 2
    l16 = zo.get_lengths(f, 15, heap16=True)
    l32 = zo.get_lengths(f, 15, heap16=False)
    assert l16[10] > l32[10]
