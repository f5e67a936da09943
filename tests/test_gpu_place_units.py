"""Where the compressed units land in the caller's buffer (zgpu_plan_run, include/zgpu.h): back to back,
nothing written beyond their end, units that do not fit left out and reported.  Plans of units above 16 KiB
write every unit straight to its final offset and stitch the 32-bit words neighbours share bytewise; plans
with small units stage and gather.  Both are checked, with every stream inflated by the oracle."""
import ctypes

import numpy as np
import pytest

import _oracle as zo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    torch.cuda.init()
    import zlib_es_amd as z
    return z, torch


def _units(kind):
    rng = np.random.default_rng(11)
    gens = ("text", "rndseq", "rnd", "runs")
    if kind == "large":  # every unit above 16 KiB: the direct path; compressed sizes are ragged
        return [zo.gen({"gen": gens[i % 4], "n": int(n), "seed": i}) for i, n in enumerate(rng.integers(17000, 90000, 96))]
    if kind == "large+tiny":  # still the direct path (one unit above 16 KiB is enough), many units inside one 32-bit word
        sizes = [20000] + [int(x) for x in rng.integers(0, 12, 400)] + [70000, 0, 0, 1, 30000]
        return [zo.gen({"gen": gens[i % 4], "n": n, "seed": i}) for i, n in enumerate(sizes)]
    if kind == "mixed":  # mostly small units by bytes, so staged; the few large ones are emitted from their own list
        sizes = [int(x) for x in rng.integers(0, 3000, 300)] + [150000] + [int(x) for x in rng.integers(9000, 16000, 40)] + [70000, 5, 0]
        return [zo.gen({"gen": gens[i % 4], "n": n, "seed": i}) for i, n in enumerate(sizes)]
    return [zo.gen({"gen": gens[i % 4], "n": int(n), "seed": i}) for i, n in enumerate(rng.integers(0, 3000, 700))]  # staged


def _run(z, torch, units, ctype, cap=None, guard=64):
    L = z._native.lib()
    sizes = np.array([u.size for u in units], dtype=np.uint64)
    in_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    data = torch.from_numpy(np.concatenate(units + [np.zeros(1, np.uint8)])).cuda()
    plan = z.Plan(in_off, ctype, 0, z._native.F_FINAL_EACH)
    room = plan.out_bound if cap is None else cap
    out = torch.full((room + guard,), 0xA5, dtype=torch.uint8, device="cuda")
    plan.run(data.data_ptr(), out.data_ptr(), room, None)
    off = np.zeros(len(units) + 1, dtype=np.uint64)
    rc = L.zgpu_plan_fetch(plan._h, off.ctypes.data, None, None, None)
    host = out.cpu().numpy()
    plan.close()
    return rc, off.astype(np.int64), host


@pytest.mark.parametrize("kind", ["large", "large+tiny", "small", "mixed"])
@pytest.mark.parametrize("ctype", [0, 1, 2])
def test_units_land_back_to_back_and_nothing_else_is_written(env, kind, ctype):
    z, torch = env
    units = _units(kind)
    rc, off, host = _run(z, torch, units, ctype)
    assert rc == 0
    assert off[0] == 0 and np.all(np.diff(off) > 0)
    total = int(off[-1])
    assert np.all(host[total:] == 0xA5), "bytes behind the last unit were written"
    for i, u in enumerate(units):
        back = zo.inflate_raw(host[off[i]:off[i + 1]].tobytes(), u.size)
        assert back == u.tobytes(), (kind, ctype, i)
    # a buffer of exactly the needed size works too, and the guard bytes behind it stay
    rc2, off2, host2 = _run(z, torch, units, ctype, cap=total)
    assert rc2 == 0 and np.array_equal(off, off2) and np.array_equal(host2[:total], host[:total])
    assert np.all(host2[total:] == 0xA5)


@pytest.mark.parametrize("kind", ["large", "small", "mixed"])
def test_units_beyond_the_capacity_are_left_out_and_reported(env, kind):
    z, torch = env
    units = _units(kind)
    rc, off, host = _run(z, torch, units, 2)
    total = int(off[-1])
    cap = (total // 2) & ~3
    rc2, off2, host2 = _run(z, torch, units, 2, cap=cap)
    assert rc2 == -4  # ZGPU_E_OUTCAP
    assert np.array_equal(off, off2), "offsets are reported for every unit, fitting or not"
    assert np.all(host2[cap:] == 0xA5), "bytes beyond the capacity were written"
    fitting = [i for i in range(len(units)) if off[i + 1] <= cap]
    assert fitting and len(fitting) < len(units)
    for i in fitting:
        assert host2[off[i]:off[i + 1]].tobytes() == host[off[i]:off[i + 1]].tobytes(), i


def test_output_pointer_must_be_word_aligned(env):
    z, torch = env
    u = zo.gen({"gen": "text", "n": 20000, "seed": 1})
    plan = z.Plan(np.array([0, u.size], dtype=np.uint64), 2, 0, 1)
    data = torch.from_numpy(u).cuda()
    out = torch.empty(plan.out_bound + 8, dtype=torch.uint8, device="cuda")
    rc = z._native.lib().zgpu_plan_run(plan._h, data.data_ptr(), out.data_ptr() + 1, plan.out_bound, None)
    plan.close()
    assert rc == -2  # ZGPU_E_ARG
