"""Streams the GPU path produced (tests/golden/gpu_streams.json, written by
tools/make_gpu_streams.py on an MI355X) against decoders that never saw the encoder:
the oracle's strict inflater everywhere, and -- in the build container, where the reference
tree and Node are present -- the reference's own RawInflate / Inflate / Gunzip / Unzip
(tests/golden/check_gpu_streams.mjs; zlib streams also through its InflateStream in pieces,
encrypted archives through Unzip with the password).  No GPU is needed: the fixture is data."""
import base64
import hashlib
import io
import json
import os
import shutil
import subprocess
import zipfile

import pytest

import _oracle as zo

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "gpu_streams.json")
REF = "/root/reference"


def _streams():
    return json.load(open(FIX))["streams"]


def test_fixture_covers_the_containers():
    kinds = {}
    for s in _streams():
        kinds[s["kind"]] = kinds.get(s["kind"], 0) + 1
    assert kinds["raw"] >= 20 and kinds["zlib"] >= 10 and kinds["gzip"] >= 10 and kinds["zip"] >= 3  # one plain, two under a password (Python mirror, JS shim)


def test_gpu_streams_decode_with_the_oracle():
    for s in _streams():
        comp = base64.b64decode(s["b64"])
        if s["kind"] == "zip" and s.get("password"):
            # encrypted entries: the cipher undone with CPython's decrypter, the twelve header bytes dropped
            # unchecked (the reference fills all twelve from Math.random, src/zip.js:144), then the strict inflater
            with zipfile.ZipFile(io.BytesIO(comp)) as zf:
                infos = {i.filename: i for i in zf.infolist()}
            for e in s["entries"]:
                i = infos[e["name"]]
                assert i.flag_bits & 1
                o = i.header_offset
                skip = 30 + int.from_bytes(comp[o + 26:o + 28], "little") + int.from_bytes(comp[o + 28:o + 30], "little")
                body = zipfile._ZipDecrypter(bytes(s["password"]))(comp[o + skip:o + skip + i.compress_size])[12:]
                plain = zo.inflate_raw(body, i.file_size) if i.compress_type == 8 else body
                assert len(plain) == e["n"] and hashlib.sha256(plain).hexdigest() == e["sha256"]
            continue
        if s["kind"] == "zip":
            with zipfile.ZipFile(io.BytesIO(comp)) as zf:
                assert zf.testzip() is None
                for e in s["entries"]:
                    assert hashlib.sha256(zf.read(e["name"])).hexdigest() == e["sha256"]
            continue
        got = {"raw": zo.inflate_raw, "zlib": zo.inflate_zlib, "gzip": zo.gunzip}[s["kind"]](comp, s["n"])
        assert len(got) == s["n"] and hashlib.sha256(got).hexdigest() == s["sha256"], (s["kind"], s.get("spec"))


@pytest.mark.skipif(shutil.which("node") is None or not os.path.isdir(os.path.join(REF, "src")),
                    reason="needs Node and the reference tree (build container only)")
def test_gpu_streams_decode_with_the_reference_decoders():
    r = subprocess.run(["node", os.path.join(HERE, "golden", "check_gpu_streams.mjs"), REF], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["ok"] == res["streams"] and not res["bad"]
    assert res["streamed"] >= 10  # zlib streams also went through the reference's InflateStream in pieces
