#!/usr/bin/env python3
"""Runs on the GPU box: compresses the named inputs of tests/_gpu_streams.py through the product
path (Python mirror of the reference classes over the C ABI, and the JavaScript shim under Node)
and writes the streams with the SHA-256 of their inputs to gpurun_out/gpu_streams.json.  Copied to
tests/golden/gpu_streams.json, the file lets the build container -- which holds the reference but
no GPU -- run every stream through the reference's own RawInflate / Inflate / Gunzip / Unzip
(tests/golden/check_gpu_streams.mjs); on a GPU, tests/test_gpu_parity_golden.py regenerates the
streams and expects the same bytes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _gpu_streams  # noqa: E402


def main():
    out = _gpu_streams.make_streams()
    path = os.path.join(ROOT, "gpurun_out", "gpu_streams.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out["streams"]), "streams")


main()
