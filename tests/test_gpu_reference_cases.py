"""The reference's compression test cases, restated against the Python mirror
of its API (zlib_es_amd.RawDeflate / Deflate / Gzip / Zip / CRC32 / Adler32),
which runs everything on the GPU.  Sources of the cases:
test/browser.es/zlib-test.js:15-84, raw-inflate-test.js:25-108,
code-path-test.js:21-31, gzip-test.js:21-76, zip-test.js:17-152.
Decoding uses the oracle's strict decoders and Python's zipfile.
"""
import io
import zipfile

import numpy as np
import pytest

import _oracle as zo

pytestmark = pytest.mark.gpu
SIZE = 76543  # zlib-test.js:7


@pytest.fixture(scope="module")
def z():
    import zlib_es_amd
    return zlib_es_amd


@pytest.fixture(scope="module")
def inputs():
    return {
        "random": zo.gen({"gen": "rnd", "n": SIZE, "seed": 20121006}),
        "sequential": zo.gen({"gen": "seq", "n": SIZE}),
        "random sequential": zo.gen({"gen": "rndseq", "n": SIZE, "seed": 1346432776267}),
    }


@pytest.mark.parametrize("ctype", [0, 1, 2])
def test_deflate_inflate_matrix(z, inputs, ctype):
    for name, data in inputs.items():
        deflated = z.Deflate(data, {"compressionType": ctype}).compress()
        assert zo.inflate_zlib(deflated, data.size) == data.tobytes(), name  # Adler-32 verified
        raw = z.RawDeflate(data, {"compressionType": ctype}).compress()
        assert zo.inflate_raw(raw, data.size) == data.tobytes(), name


def test_undercomitted(z):
    data = list(range(1, 24))
    for out in (z.Deflate(data).compress(), z.Deflate.compress(data)):
        assert zo.inflate_zlib(out, 23) == bytes(data)


def test_zlib_header_bytes_match_reference(z, golden):
    # CMF/FLG and the Adler-32 trailer are byte-identical to the reference's (deflate.js:49-97)
    for c in golden["cases"]:
        if not c["n"] or c["n"] > 70000:
            continue
        data = zo.gen(c["spec"])
        for ct in (0, 1, 2):
            ours = bytes(z.Deflate(data, {"compressionType": ct}).compress())
            ref = zo.zlib_compress(data, ct)
            assert ours[:2] == ref[:2] and ours[-4:] == ref[-4:]
            if ct == 0:
                assert ours == ref  # stored framing is fully determined


def test_code_path(z, inputs, monkeypatch):
    # code-path-test.js:21-31: exactly the requested block maker runs
    names = ["makeNocompressBlock", "makeFixedHuffmanBlock", "makeDynamicHuffmanBlock"]
    calls = {n: 0 for n in names}
    for n in names:
        orig = getattr(z.RawDeflate, n)

        def spy(self, block, final, _n=n, _orig=orig):
            calls[_n] += 1
            return _orig(self, block, final)
        monkeypatch.setattr(z.RawDeflate, n, spy)
    for ctype, want in enumerate(names):
        for n in names:
            calls[n] = 0
        out = z.Deflate(inputs["sequential"], {"compressionType": ctype}).compress()
        assert [calls[n] > 0 for n in names] == [n == want for n in names]
        assert zo.inflate_zlib(out, SIZE) == inputs["sequential"].tobytes()
    with pytest.raises(ValueError):
        z.RawDeflate(inputs["sequential"], {"compressionType": 3}).compress()


def test_output_buffer_and_index(z):
    # outputBuffer/outputIndex: the caller's header bytes stay in front (rawdeflate.js:89-94)
    data = zo.gen({"gen": "text", "n": 5000, "seed": 1})
    buf = np.zeros(64, dtype=np.uint8)
    buf[:5] = [9, 8, 7, 6, 5]
    rd = z.RawDeflate(data, {"outputBuffer": buf, "outputIndex": 5, "compressionType": 2})
    out = rd.compress()
    assert out[:5].tolist() == [9, 8, 7, 6, 5] and rd.op == out.size
    assert zo.inflate_raw(out[5:], data.size) == data.tobytes()


def test_gzip(z, inputs, golden):
    data = inputs["random sequential"]
    assert zo.gunzip(z.Gzip(data).compress(), data.size) == data.tobytes()
    g = z.Gzip(data, {"flags": {"fname": True, "fhcrc": True}, "filename": "foobar.filename"})
    g.mtime = 1346432776
    out = bytes(g.compress())
    assert zo.gunzip(out, data.size) == data.tobytes()
    # header and trailer byte-identical to the reference's output for the same options (golden)
    ref = next(x for x in golden["gzip"] if x["opt"].get("filename") == "foobar.filename")
    assert ref["out"]["len"] > 0
    hdr_len = 10 + len("foobar.filename") + 1 + 2
    refbytes = zo.gzip_compress(data, 2, 0, ref["mtime"], 1 | 4, "foobar.filename")
    assert out[:hdr_len] == refbytes[:hdr_len] and out[-8:] == refbytes[-8:]


def test_zip_against_reference_framing(z, golden):
    gz = golden["zip"][0]
    ref = bytes.fromhex(gz["out"]["hex"])
    zp = z.Zip({"comment": gz["comment"].encode()})
    plains = {}
    for f in gz["files"]:
        data = zo.gen(f["spec"])
        plains[f["name"]] = data.tobytes()
        opt = {"filename": f["name"].encode(), "compressionMethod": f["method"], "date": tuple(gz["date"])}
        if "comment" in f:
            opt["comment"] = f["comment"].encode()
        if "ctype" in f:
            opt["deflateOption"] = {"compressionType": f["ctype"]}
        zp.addFile(data, opt)
    ours = bytes(zp.compress())
    a, b = zipfile.ZipFile(io.BytesIO(ours)), zipfile.ZipFile(io.BytesIO(ref))
    assert a.testzip() is None  # every CRC-32 checks out
    assert a.comment == b.comment
    for ia, ib in zip(a.infolist(), b.infolist()):
        for field in ("filename", "compress_type", "CRC", "file_size", "date_time", "comment", "flag_bits",
                      "create_version", "extract_version", "create_system", "external_attr", "internal_attr"):
            assert getattr(ia, field) == getattr(ib, field), field
        assert a.read(ia) == plains[ia.filename]
    # the STORE entry and all framing around it are byte-identical
    n0 = 30 + len("hoge.txt") + 5
    assert ours[:n0] == ref[:n0]


def test_zip_many_small_files(z):
    rng = np.random.default_rng(5)
    zp = z.Zip()
    plains = {}
    for i in range(400):
        n = int(rng.integers(1, 16385))
        data = zo.gen({"gen": ["text", "rndseq", "rnd", "runs"][i % 4], "n": n, "seed": i})
        plains["f%d" % i] = data.tobytes()
        zp.addFile(data, {"filename": ("f%d" % i).encode(), "date": (2020, 1, 2, 3, 4, 6),
                          "deflateOption": {"compressionType": 1 + (i & 1)}})
    zf = zipfile.ZipFile(io.BytesIO(bytes(zp.compress())))
    assert zf.testzip() is None and len(zf.namelist()) == 400
    for name, plain in plains.items():
        assert zf.read(name) == plain


def test_gzip_chunked_single_member(z):
    data = zo.gen({"gen": "text", "n": 3 * (1 << 20) + 4321, "seed": 8})
    gzb = z.gzip_chunked(data, chunk_size=1 << 20, mtime=7)
    assert zo.gunzip(gzb, data.size) == data.tobytes()  # CRC-32 (by combine) and ISIZE verified
    import gzip as pygzip
    assert pygzip.decompress(gzb) == data.tobytes()


def _lcg():
    st = [12345]

    def rnd():  # the generator tests/golden/gen_golden.mjs pins Math.random to
        st[0] = st[0] * 48271 % 2147483647
        return st[0] / 2147483647
    return rnd


def _read_encrypted(archive, passwords):
    """{name: plain bytes} of an archive whose entries are encrypted; the twelve header bytes are
    dropped unchecked, as the reference's Unzip does (src/unzip.js:113-117)."""
    out = {}
    zf = zipfile.ZipFile(io.BytesIO(archive))
    for info in zf.infolist():
        assert info.flag_bits & 1
        o = info.header_offset
        nlen, xlen = int.from_bytes(archive[o + 26:o + 28], "little"), int.from_bytes(archive[o + 28:o + 30], "little")
        body = archive[o + 30 + nlen + xlen:o + 30 + nlen + xlen + info.compress_size]
        dec = zipfile._ZipDecrypter(bytes(passwords[info.filename]))(body)[12:]
        plain = zo.inflate_raw(dec, info.file_size) if info.compress_type == 8 else dec
        assert zo.crc32(np.frombuffer(plain, dtype=np.uint8)) == info.CRC
        out[info.filename] = plain
    return out


def test_zip_with_password_equals_the_reference_archive(z, golden):
    # STORE entries under a password, header bytes from the same pinned generator: the archive is the
    # reference's byte for byte (src/zip.js:138-147,177-180, incl. the file-index test at :144)
    g = golden["zipcrypto_archive"]
    zp = z.Zip()
    zp.random = _lcg()
    for f in g["files"]:
        opt = {"filename": f["name"].encode(), "compressionMethod": 0, "date": tuple(g["date"])}
        if f["password"] is not None:
            opt["password"] = f["password"]
        zp.addFile(zo.gen(f["spec"]), opt)
    zp.setPassword(g["password"])
    assert bytes(zp.compress()).hex() == g["out"]["hex"]


def test_zip_compress_with_password(z):
    # test/browser.es/zip-test.js:83-150: one password for the archive, then one per file
    datas = {"hogehoge": zo.gen({"gen": "text", "n": 30000, "seed": 1}), "fugafuga": zo.gen({"gen": "rndseq", "n": 9000, "seed": 2}),
             "piyopiyo": zo.gen({"gen": "rnd", "n": 700, "seed": 3})}
    pw = list(range(10))
    zp = z.Zip()
    zp.setPassword(pw)
    for name, d in datas.items():
        zp.addFile(d, {"filename": name.encode(), "date": (2020, 1, 2, 3, 4, 6)})
    got = _read_encrypted(bytes(zp.compress()), {name: pw for name in datas})
    assert got == {name: d.tobytes() for name, d in datas.items()}
    each = {name: [i + 1, 7, len(name)] for i, name in enumerate(datas)}
    zp = z.Zip()
    for name, d in datas.items():
        zp.addFile(d, {"filename": name.encode(), "password": each[name], "date": (2020, 1, 2, 3, 4, 6),
                       "deflateOption": {"compressionType": 1}})
    got = _read_encrypted(bytes(zp.compress()), each)
    assert got == {name: d.tobytes() for name, d in datas.items()}


def test_zip_password_as_a_byte_array(z):
    # the reference's passwords are Uint8Arrays (src/zip.js:95,287-294); their Python counterpart is a numpy array
    d = zo.gen({"gen": "text", "n": 12000, "seed": 8})
    pw_archive, pw_file = np.arange(9, dtype=np.uint8) + 3, np.array([200, 1, 2, 3], dtype=np.uint8)
    zp = z.Zip()
    zp.setPassword(pw_archive)
    zp.addFile(d, {"filename": b"whole", "date": (2020, 1, 2, 3, 4, 6)})
    zp.addFile(d[:5000], {"filename": b"own", "password": pw_file, "date": (2020, 1, 2, 3, 4, 6)})
    got = _read_encrypted(bytes(zp.compress()), {"whole": pw_archive.tolist(), "own": pw_file.tolist()})
    assert got == {"whole": d.tobytes(), "own": d[:5000].tobytes()}
