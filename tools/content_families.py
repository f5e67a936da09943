#!/usr/bin/env python3
"""Size against the reference's algorithm for content families beyond the named test cases
(skewed and structured small alphabets, digits, hex, base64, short runs, Markov sources):
one 64 KiB unit of each, FIXED and DYNAMIC, at the default settings.  Which of them the
dense-class pass took is seen by running them again with the pass off."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as zo  # noqa: E402
import zlib_es_amd as z  # noqa: E402

N = int(os.environ.get("N", "65536"))
rng = np.random.default_rng(11)


def markov(k, stay):
    out = np.empty(N, dtype=np.uint8)
    v = 0
    r = rng.random(N)
    nxt = rng.integers(0, k, N)
    for i in range(N):
        if r[i] >= stay:
            v = nxt[i]
        out[i] = v
    return out


def runs_of(k, maxlen):
    parts = []
    total = 0
    while total < N:
        n = int(rng.integers(1, maxlen + 1))
        parts.append(np.full(n, (int(rng.integers(0, k)) * 37 + 1) & 255, dtype=np.uint8))
        total += n
    return np.concatenate(parts)[:N]


def words(alphabet, nwords, wl):
    vocab = [bytes(rng.choice(list(alphabet), size=int(rng.integers(wl[0], wl[1] + 1))).astype(np.uint8)) for _ in range(nwords)]
    out = bytearray()
    while len(out) < N:
        out += vocab[int(rng.integers(0, nwords))]
    return np.frombuffer(bytes(out[:N]), dtype=np.uint8)


FAMILIES = {
    "geometric p=.5": lambda: np.minimum(rng.geometric(0.5, N) - 1, 255).astype(np.uint8),
    "geometric p=.3": lambda: np.minimum(rng.geometric(0.3, N) - 1, 255).astype(np.uint8),
    "geometric p=.15": lambda: np.minimum(rng.geometric(0.15, N) - 1, 255).astype(np.uint8),
    "zipf 16 values": lambda: rng.choice(16, N, p=(1 / np.arange(1, 17)) / (1 / np.arange(1, 17)).sum()).astype(np.uint8),
    "2 values 90:10": lambda: (rng.random(N) < 0.1).astype(np.uint8),
    "3 values 80:15:5": lambda: rng.choice(3, N, p=[0.8, 0.15, 0.05]).astype(np.uint8),
    "8 values 60:20:...": lambda: rng.choice(8, N, p=[0.6, 0.2, 0.08, 0.05, 0.03, 0.02, 0.01, 0.01]).astype(np.uint8),
    "digits": lambda: (rng.integers(0, 10, N) + 48).astype(np.uint8),
    "digits+sep csv": lambda: np.frombuffer((",".join(str(int(x)) for x in rng.integers(0, 100000, N // 4))).encode()[:N], dtype=np.uint8),
    "hex": lambda: np.frombuffer(rng.bytes(N // 2).hex().encode(), dtype=np.uint8),
    "base64": lambda: np.frombuffer(__import__("base64").b64encode(rng.bytes(N))[:N], dtype=np.uint8),
    "markov 2 stay .7": lambda: markov(2, 0.7),
    "markov 4 stay .5": lambda: markov(4, 0.5),
    "markov 4 stay .8": lambda: markov(4, 0.8),
    "runs<=4 of 4": lambda: runs_of(4, 4),
    "runs<=8 of 3": lambda: runs_of(3, 8),
    "runs<=16 of 6": lambda: runs_of(6, 16),
    "acgt words": lambda: words(b"ACGT", 300, (6, 14)),
    "acgt 2k words": lambda: words(b"ACGT", 2000, (4, 9)),
    "01 words": lambda: words(b"01", 200, (5, 12)),
    "16-bit small": lambda: np.stack([rng.integers(0, 6, N // 2).astype(np.uint8), np.zeros(N // 2, dtype=np.uint8)], 1).reshape(-1),
    "16-bit noise lo": lambda: np.stack([rng.integers(0, 256, N // 2).astype(np.uint8), rng.integers(0, 2, N // 2).astype(np.uint8)], 1).reshape(-1),
}

if "RUNS" in os.environ:  # runs of 1..R bytes of k values, for R and k from the environment ("16,32,64 3,4")
    rs, ks = [[int(x) for x in part.split(",")] for part in os.environ["RUNS"].split()]
    FAMILIES = {"runs<=%d of %d" % (r, k): (lambda r=r, k=k: runs_of(k, r)) for r in rs for k in ks}
names = list(FAMILIES)
datas = [np.ascontiguousarray(FAMILIES[k]()) for k in names]
ref = {ct: [len(zo.rawdeflate(d, ct)) for d in datas] for ct in (1, 2)}
got = {}
DENSE = int(os.environ.get("DENSE", str(z._native.DEFAULT_DENSE_CHAIN)))
ms = []
z._native.set_dense_search(DENSE)
for d in datas:  # one unit alone: what a caller waits for it (second call)
    z.deflate_batch([d], 1, 0)
    t0 = time.perf_counter()
    z.deflate_batch([d], 1, 0)
    ms.append((time.perf_counter() - t0) * 1e3)
for pas in (1, 0):
    z._native.set_dense_search(DENSE if pas else 0)
    for ct in (1, 2):
        outs, _, _ = z.deflate_batch(datas, ct, 0)
        for d, o in zip(datas, outs):
            assert zo.inflate_raw(o, d.size) == d.tobytes()
        got[pas, ct] = [len(o) for o in outs]
z._native.set_dense_search()
print("dense-class chain budget %d" % DENSE)
print("%-20s %8s %8s %9s %8s | without the dense-class pass" % ("family", "FIXED", "DYNAMIC", "out/in", "ms/unit"))
for i, k in enumerate(names):
    took = got[1, 1][i] != got[0, 1][i] or got[1, 2][i] != got[0, 2][i]
    print("%-20s %8.3f %8.3f %9.4f %8.2f | %8.3f %8.3f %s" % (
        k, got[1, 1][i] / ref[1][i], got[1, 2][i] / ref[2][i], got[1, 2][i] / datas[i].size, ms[i],
        got[0, 1][i] / ref[1][i], got[0, 2][i] / ref[2][i], "(dense class)" if took else ""), flush=True)
