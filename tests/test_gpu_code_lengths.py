"""Cost of the GPU's Huffman code lengths against the reference's.  The emit kernel builds plain
Huffman trees and repairs the ones deeper than the limit (15 bits, 7 for its code-length code);
the reference runs reverse package-merge (src/rawdeflate.js:474-574), which is optimal under the
limit (and limits distance codes to 7 bits, where the GPU may use 15).  Round trips cannot see a
silent loss here, so every dynamic block's code lengths are read back from its header and weighed
with the block's own symbol counts against the oracle's package-merge on the same counts."""
import numpy as np
import pytest

import _oracle as zo
from test_gpu_fuzz import _skewed_unit
from test_gpu_parity import _planted

pytestmark = pytest.mark.gpu


def _cost(freq, lens):
    return int((freq.astype(np.int64) * lens.astype(np.int64)).sum())


def _check(outs, datas, slack):
    worst = 0.0
    nblocks = 0
    for o, d in zip(outs, datas):
        for b in zo.block_stats(o, d.size):
            nblocks += 1
            for freq, lens, limit in ((b["litlen_freq"][:286], b["litlen_len"][:286], 15),
                                      (b["dist_freq"][:30], b["dist_len"][:30], 15)):
                used = freq > 0
                if used.sum() < 2:
                    continue  # one symbol: one bit either way
                assert (lens[used] > 0).all() and lens.max() <= limit
                assert abs(float((2.0 ** -lens[lens > 0].astype(np.float64)).sum()) - 1.0) < 1e-9  # complete code
                gpu = _cost(freq, lens)
                ref = _cost(freq, zo.get_lengths(freq, limit, heap16=False))  # package-merge, optimal under the limit
                assert gpu >= ref  # nothing beats the optimum: the bookkeeping of this test is right
                worst = max(worst, gpu / ref - 1.0)
                assert gpu <= ref * (1.0 + slack) + 2, (gpu, ref, int(used.sum()), int(lens.max()))
    return nblocks, worst


def test_code_length_cost_on_skewed_blocks():
    import zlib_es_amd as z
    rng = np.random.default_rng(21)
    datas = [_skewed_unit(rng) for _ in range(120)]
    # Fibonacci-like length counts push the plain Huffman tree past 15 bits: the repair runs
    tails = ([1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233], [2, 2, 4, 6, 10, 16, 26, 42, 68, 110, 178, 240])
    datas += [_planted(rng, 65536, t) for t in tails]
    datas += [zo.gen({"gen": "text", "n": 65536, "seed": 3}), zo.gen({"gen": "rndseq", "n": 65536, "seed": 4})]
    outs, _, _ = z.deflate_batch(datas, 2, 0)
    nblocks, worst = _check(outs, datas, 0.002)
    assert nblocks >= len(datas)
    print("blocks %d, worst excess over package-merge %.4f %%" % (nblocks, worst * 100))
