"""Extracts the decode-only known-answer vectors held by the reference's own
tests (test/browser.es/inflate-test.js, gunzip-test.js) into
tests/golden/decode_fixtures.json.  Only data (base64 streams and the expected
plain bytes) is taken; run in the build container where /root/reference is
mounted.
"""
import base64
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def b64s(block):
    return "".join(re.findall(r"'([A-Za-z0-9+/=]+)'", block))


src = open(os.path.join(REF, "test/browser.es/inflate-test.js")).read()
out = {"zlib": [], "gzip": []}
m = re.search(r"const fixedData =(.*?);", src, re.S)
fixed = base64.b64decode(b64s(m.group(1)))
assert len(fixed) == 1202
out["zlib"].append({"name": "pre-deflated seq(123456)", "stream_hex": fixed.hex(), "plain": {"gen": "seq", "n": 123456}})
for m in re.finditer(r"it\('(issue#35[^']*)'.*?base64toArray\((.*?)\);\s*const plain = \[(.*?)\];", src, re.S):
    stream = base64.b64decode(b64s(m.group(2)))
    plain = bytes(int(x) for x in re.findall(r"\d+", m.group(3)))
    out["zlib"].append({"name": m.group(1), "stream_hex": stream.hex(), "plain": {"gen": "bytes", "hex": plain.hex()}})
src = open(os.path.join(REF, "test/browser.es/gunzip-test.js")).read()
for name, b64, plain in (("aaaa", "H4sIAAAAAAAAA0tMTEwEAEXlmK0EAAAA", b"aaaa"),
                         ("hello world with filename", "H4sICOzl1k8AA2hvZ2UudHh0AMtIzcnJVyjPL8pJ4QIALTsIrwwAAAA=", b"hello world\n")):
    assert b64 in src
    out["gzip"].append({"name": name, "stream_hex": base64.b64decode(b64).hex(), "plain": {"gen": "bytes", "hex": plain.hex()}})
json.dump(out, open(os.path.join(HERE, "decode_fixtures.json"), "w"))
print({k: len(v) for k, v in out.items()})
