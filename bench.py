#!/usr/bin/env python3
"""Benchmark of the GPU DEFLATE hot path (one JSON line on rank 0).

A step = one pass of the batched RawDeflate path over one batch of synthetic
input resident in HBM: by default 65536 independent 64 KiB units of mixed
content (50 % word text, 25 % structured binary, 15 % random, 10 % low
entropy), compressionType DYNAMIC, lazy 0.  With --gpus N (launched through
torch.distributed.run) every rank compresses its own batch of that size
(weak scaling, no data-path collective); value = all ranks' input bytes
divided by the slowest rank's time.

The CPU oracle appears here only as the checker and as the `cpu_baseline`
leg (rank 0, N = 1, bounded sample); it is never what is timed as `value`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "deflate input GB/s (node, DYNAMIC) at 1/2/4/8 GPUs + ratio vs ref"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


# --------------------------------------------------------------------------
# synthetic units, generated on the GPU
# --------------------------------------------------------------------------
def _vocab(dev):
    rng = np.random.default_rng(0x5EED)
    lens = rng.integers(2, 11, 4096)
    tab = np.full((4096, 12), 32, dtype=np.uint8)
    for w in range(4096):
        tab[w, : lens[w]] = rng.integers(97, 123, lens[w])
    return torch.from_numpy(tab).to(dev), torch.from_numpy((lens + 1).astype(np.int64)).to(dev)


def _from_runs(run_len, value_fn, ub, gen):
    """rows of variable-length runs -> [nb, ub] bytes via cumsum + searchsorted"""
    csum = torch.cumsum(run_len, dim=1)
    pos = torch.arange(ub, device=run_len.device).expand(run_len.shape[0], ub).contiguous()
    j = torch.searchsorted(csum, pos, right=True).clamp_(max=run_len.shape[1] - 1)
    start = torch.gather(csum, 1, j) - torch.gather(run_len, 1, j)
    return value_fn(j, pos - start)


def gen_units(n_units, ub, seed, dev):
    """uint8 tensor [n_units * ub]; unit kind by index mod 20: 10 text, 5 binary, 3 random, 2 low entropy"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = torch.empty((n_units, ub), dtype=torch.uint8, device=dev)
    kind = torch.arange(n_units, device=dev) % 20
    vocab, vlen = _vocab(dev)
    step = 1024
    for lo in range(0, n_units, step):
        hi = min(n_units, lo + step)
        k = kind[lo:hi]
        blk = out[lo:hi]
        # text
        idx = (k < 10).nonzero().flatten()
        if idx.numel():
            nw = ub // 3 + 8
            ids = (torch.rand((idx.numel(), nw), device=dev, generator=g).pow_(3) * 4096).long().clamp_(max=4095)
            wl = vlen[ids]
            blk[idx] = _from_runs(wl, lambda j, off: vocab[torch.gather(ids, 1, j), off], ub, g)
        # structured binary: runs of incrementing bytes (random start, random length)
        idx = ((k >= 10) & (k < 15)).nonzero().flatten()
        if idx.numel():
            nr = ub // 24 + 8
            rl = torch.randint(1, 257, (idx.numel(), nr), device=dev, generator=g)
            st = torch.randint(0, 256, (idx.numel(), nr), device=dev, generator=g)
            blk[idx] = _from_runs(rl, lambda j, off: ((torch.gather(st, 1, j) + off) & 255).to(torch.uint8), ub, g)
        # random
        idx = ((k >= 15) & (k < 18)).nonzero().flatten()
        if idx.numel():
            blk[idx] = torch.randint(0, 256, (idx.numel(), ub), device=dev, generator=g, dtype=torch.uint8)
        # low entropy: long runs of four byte values
        idx = (k >= 18).nonzero().flatten()
        if idx.numel():
            nr = ub // 48 + 8
            rl = torch.randint(1, 513, (idx.numel(), nr), device=dev, generator=g)
            vv = torch.randint(0, 4, (idx.numel(), nr), device=dev, generator=g) * 85
            blk[idx] = _from_runs(rl, lambda j, off: torch.gather(vv, 1, j).to(torch.uint8), ub, g)
    return out.view(-1)


# --------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--units", type=int, default=65536)
    ap.add_argument("--unit-bytes", type=int, default=65536)
    ap.add_argument("--ctype", type=int, default=2)
    ap.add_argument("--lazy", type=int, default=0)
    ap.add_argument("--workload", default="deflate", choices=["deflate", "checksum"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--chain", type=int, default=0, help="hash-chain search depth (0 = library default)")
    ap.add_argument("--nice", type=int, default=32)
    ap.add_argument("--deep", type=int, default=-1, help="long-match deep-walk budget (-1 = library default)")
    ap.add_argument("--fan", type=int, default=-1, help="probe-ahead fans per 64 bytes (-1 = library default)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path to time)")
    # one rank per GPU; ZGPU_BENCH_BACKEND=gloo rehearses the multi-rank flow on a box with
    # fewer GPUs than ranks (ranks then share devices and the reductions run on the CPU)
    backend = os.environ.get("ZGPU_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit("rank %d has no GPU of its own (%d visible)" % (local_rank, ndev))
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    import zlib_es_amd as z
    N = z._native
    if args.chain or args.nice != 32:
        z.set_search(args.chain or z._native.DEFAULT_MAX_CHAIN, args.nice)

    if args.deep >= 0 or args.fan >= 0:
        N.set_long_match(args.deep if args.deep >= 0 else N.DEFAULT_DEEP_CHAIN,
                         args.fan if args.fan >= 0 else N.DEFAULT_PROBE_FANS)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "checksum":
        return bench_checksum(args, z, dev, rank, world, dist, barrier)

    nu, ub = args.units, args.unit_bytes
    data = gen_units(nu, ub, 1234 + rank, dev)
    in_off = np.arange(nu + 1, dtype=np.uint64) * np.uint64(ub)
    plan = z.Plan(in_off, args.ctype, args.lazy, N.F_FINAL_EACH)
    out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    for _ in range(args.warmup):
        plan.run(data.data_ptr(), out.data_ptr(), out.numel(), stream)
    barrier()
    t0 = time.perf_counter()
    kern_ms = []
    for _ in range(args.steps):
        plan.run(data.data_ptr(), out.data_ptr(), out.numel(), stream)
        kern_ms.append(None)  # events are read after the timed region
    barrier()
    dt = time.perf_counter() - t0
    # the library's HIP events bracket the kernels of the last run on this stream
    k_ms, tot_ms = plan.kernel_ms()
    search_ms, emit_ms = plan.phase_ms()
    out_off, _, _ = plan.fetch()
    out_bytes = int(out_off[-1])
    in_bytes = nu * ub

    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    tot_in = torch.tensor([float(in_bytes)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot_in, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    value = float(tot_in.item()) * args.steps / dt / 1e9

    res = {
        "metric": METRIC, "value": round(value, 3), "unit": "GB/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "batched RawDeflate, %d x %d B independent units of mixed text/binary per GPU, "
                               "compressionType %d, lazy %d" % (nu, ub, args.ctype, args.lazy),
                   "units_per_gpu": nu, "unit_bytes": ub, "parallelism": "units sharded over %d GPU(s), no collective" % world},
        "ratio_out_over_in": round(out_bytes / in_bytes, 5),
    }
    if rank == 0:
        # read the input once, write the compressed stream once; the path is two kernels
        # (match search, then Huffman coding + bit emission), priced together
        algo_bytes = in_bytes + out_bytes
        res["roofline"] = {"bound": "hbm", "achieved": round(algo_bytes / (k_ms * 1e-3) / 1e9, 3),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(algo_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                           "traffic": measured_traffic(nu, ub, args),
                           "kernel": "zgpu_deflate_kernel + zgpu_emit_kernel", "kernel_ms": round(k_ms, 3),
                           "search_kernel_ms": round(search_ms, 3), "emit_kernel_ms": round(emit_ms, 3)}
        lds = measured_lds(nu, ub, args)
        if lds:
            res["search_kernel_occupancy"] = lds  # what actually bounds the path: VALU issue and the LDS pipe
        res.update(verify_all(data, in_off, out, out_off, dev))
        if world == 1 and not args.no_cpu:
            res.update(cpu_leg(args, data, out, out_off, nu, ub))
        print(json.dumps(res), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


def measured_traffic(nu, ub, args):
    """HBM-side bytes per launch from the committed rocprofv3 counter pass of this exact
    workload (profiles/traffic.json, written by tools/summarize_profiles.py), else None."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None
    w = t.get("workload", {})
    if (w.get("units"), w.get("unit_bytes"), w.get("ctype"), w.get("lazy")) != (nu, ub, args.ctype, args.lazy):
        return None
    return t.get("bytes_per_launch")


def measured_lds(nu, ub, args):
    """LDS-pipe and VALU occupancy of the search kernel from the same committed counter passes
    (measured on 8192 units of this mix), or None for another workload."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None
    w = t.get("workload", {})
    if (w.get("unit_bytes"), w.get("ctype"), w.get("lazy")) != (ub, args.ctype, args.lazy):
        return None
    return t.get("lds")


def verify_all(data, in_off, out, out_off, dev):
    """Every unit of the batch through the GPU round-trip verifier (a strict DEFLATE decoder
    that compares with the input as it goes; test infrastructure, outside the timed region)."""
    try:
        import _gpu_verify as gv
        t0 = time.perf_counter()
        st = gv.verify(out, torch.from_numpy(out_off.astype(np.int64)).to(dev), data,
                       torch.from_numpy(in_off.astype(np.int64)).to(dev), True)
        bad = int((st != 0).sum())
        return {"roundtrip_all_units": bad == 0, "units_verified_on_gpu": int(st.numel()), "units_failed": bad,
                "verify_ms": round((time.perf_counter() - t0) * 1e3, 1)}
    except OSError as e:  # the verifier library is optional for the timing itself
        return {"roundtrip_all_units": None, "verify_error": str(e)}


def cpu_leg(args, data, out, out_off, nu, ub):
    """Oracle on a bounded sample of the same units: baseline speed, size ratio, round trip."""
    import _oracle as zo
    stride = max(1, nu // 4096)
    t_used, n_done, ref_bytes, gpu_bytes, in_b = 0.0, 0, 0, 0, 0
    ok = True
    for u in range(0, nu, stride):
        if t_used > args.cpu_seconds:
            break
        unit = data[u * ub:(u + 1) * ub].cpu().numpy()
        comp = out[int(out_off[u]):int(out_off[u + 1])].cpu().numpy()
        t0 = time.perf_counter()
        ref = zo.rawdeflate(unit, args.ctype, args.lazy)
        t_used += time.perf_counter() - t0
        ok = ok and zo.inflate_raw(comp, ub) == unit.tobytes()
        ref_bytes += len(ref)
        gpu_bytes += comp.size
        in_b += ub
        n_done += 1
    return {
        "cpu_baseline": {"value": round(in_b / t_used / 1e9, 5), "unit": "GB/s", "cores": 1, "kind": "port",
                         "sample": "%d of the same %d-byte units (every %dth), C restatement of "
                                   "src/rawdeflate.js, one thread" % (n_done, ub, stride)},
        "ratio_vs_ref": round(gpu_bytes / ref_bytes, 5), "roundtrip_ok": bool(ok), "sample_units": n_done,
    }


def bench_checksum(args, z, dev, rank, world, dist, barrier):
    nbytes = args.units * args.unit_bytes
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    data = torch.randint(0, 256, (nbytes,), device=dev, generator=g, dtype=torch.uint8)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(args.warmup):
        z._native.checksum_device(data.data_ptr(), nbytes, stream)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        crc, ad = z._native.checksum_device(data.data_ptr(), nbytes, stream)
    barrier()
    dt = time.perf_counter() - t0
    k_ms = z._native.checksum_kernel_ms()
    res = {"metric": "CRC32 + Adler32 input GB/s", "value": round(nbytes * world * args.steps / dt / 1e9, 3),
           "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "CRC32 + Adler32 over %d random bytes per GPU" % nbytes},
           "roofline": {"bound": "hbm", "achieved": round(nbytes / (k_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(nbytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "traffic": None,
                        "kernel": "zgpu_checksum_wide_kernel" if nbytes >= (64 << 20) else "zgpu_checksum_kernel",
                        "kernel_ms": round(k_ms, 3)}}
    if rank == 0 and world == 1 and not args.no_cpu:
        import _oracle as zo
        n = min(nbytes, 256 << 20)
        host = data[:n].cpu().numpy()
        t0 = time.perf_counter()
        c, a = zo.crc32(host), zo.adler32(host)
        t = time.perf_counter() - t0
        gc, ga = z._native.checksum_device(data.data_ptr(), n, stream)
        res["cpu_baseline"] = {"value": round(n / t / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                               "sample": "first %d bytes, C restatement of src/crc32.js + src/adler32.js" % n}
        res["bit_exact"] = bool((c, a) == (gc, ga))
    if rank == 0:
        print(json.dumps(res), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
