#!/usr/bin/env python3
"""Benchmark of the GPU DEFLATE hot path (one JSON line on rank 0).

A step = one pass of the batched RawDeflate path over one batch of synthetic
input resident in HBM.  Workloads (--workload):

  deflate   (default; BASELINE config 3) 65536 independent 64 KiB units of mixed content
            (50 % word text, 25 % structured binary, 15 % random, 10 % low entropy),
            compressionType DYNAMIC, lazy 0
  zip       (config 4) files of 1-16 KiB, uniform, same content mix, half FIXED half DYNAMIC,
            CRC-32 per file: what Zip.compress()'s per-file loop turns into
  gzip      (config 5) 1 MiB chunks of one stream per GiB (each chunk primed with the 32 KiB
            before it, byte-aligned end), CRC-32 per chunk, combined on the host
  checksum  (config 2) CRC-32 + Adler-32 over one buffer
  fewvalues (no BASELINE configuration) 64 KiB units of two to six equally likely byte values: what the
            library's dense-class pass is for

With --gpus N (launched through torch.distributed.run) every rank compresses
its own batch of that size (weak scaling, no data-path collective); value =
all ranks' input bytes divided by the slowest rank's time.

The CPU oracle appears here only as the checker and as the `cpu_baseline`
leg (rank 0, N = 1, bounded sample); it is never what is timed as `value`.
"""
import argparse
import ctypes
import json
import os
import shutil
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

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
    step = max(1, (1024 * 65536) // ub)
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


def gen_files(n_files, max_bytes, seed, dev):
    """files of 1..max_bytes bytes (uniform), content as gen_units, packed back to back:
    (flat uint8 tensor, in_off uint64 numpy array of n_files + 1 offsets)"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed ^ 0xF11E)
    sizes = torch.randint(1, max_bytes + 1, (n_files,), device=dev, generator=g)
    parts = []
    step = 32768
    for lo in range(0, n_files, step):
        hi = min(n_files, lo + step)
        rows = gen_units(hi - lo, max_bytes, seed + lo, dev).view(hi - lo, max_bytes)
        keep = torch.arange(max_bytes, device=dev).unsqueeze(0) < sizes[lo:hi].unsqueeze(1)
        parts.append(rows[keep])
    off = np.zeros(n_files + 1, dtype=np.uint64)
    off[1:] = np.cumsum(sizes.cpu().numpy().astype(np.uint64))
    return torch.cat(parts), off


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        n = min(n, int(env))  # the box's CPU share
    return max(1, n)


# --------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--units", type=int, default=0, help="units / files / chunks per GPU (0 = the workload's default)")
    ap.add_argument("--unit-bytes", type=int, default=0, help="unit size (deflate), largest file (zip), chunk (gzip)")
    ap.add_argument("--ctype", type=int, default=2)
    ap.add_argument("--lazy", type=int, default=0)
    ap.add_argument("--workload", default="deflate", choices=["deflate", "zip", "gzip", "checksum", "fewvalues"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-path and secondary-workload legs")
    ap.add_argument("--chain", type=int, default=0, help="hash-chain search depth (0 = library default)")
    ap.add_argument("--nice", type=int, default=32)
    ap.add_argument("--deep", type=int, default=-1, help="long-match deep-walk budget (-1 = library default)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path to time)")
    # One rank per GPU.  The ranks exchange no data (units are independent): the process group carries
    # the barriers and two scalar reductions of the timing, which run over gloo on the CPU -- RCCL is
    # not part of this path.  ZGPU_BENCH_SHARE_GPUS=1 rehearses the multi-rank flow on a box with
    # fewer GPUs than ranks (ranks then share devices).
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and os.environ.get("ZGPU_BENCH_SHARE_GPUS") != "1":
        raise SystemExit("rank %d has no GPU of its own (%d visible)" % (local_rank, ndev))
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("gloo")
    red_dev = torch.device("cpu")

    import zlib_es_amd as z
    N = z._native
    if args.chain or args.nice != 32:
        z.set_search(args.chain or z._native.DEFAULT_MAX_CHAIN, args.nice)
    if args.deep >= 0:
        N.set_long_match(args.deep)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ctx = dict(args=args, z=z, N=N, dev=dev, rank=rank, world=world, dist=dist, barrier=barrier, red_dev=red_dev)
    if args.workload == "checksum":
        res = bench_checksum(ctx)
    else:
        res = bench_deflate(ctx, args.workload)
        if rank == 0 and world == 1 and args.workload == "deflate" and not args.no_extras:
            # the other configurations and the drop-in path, in short form, with the same line shape
            legs = {"zip": lambda: brief(bench_deflate(ctx, "zip", short=True)),
                    "gzip": lambda: brief(bench_deflate(ctx, "gzip", short=True)),
                    # the same 65536 x 64 KiB batch under FIXED coding (BASELINE config 3 names both modes)
                    "fixed": lambda: brief(bench_deflate(ctx, "deflate", short=True, ctype=1)),
                    # BASELINE config 2: CRC-32 + Adler-32 over 4 GiB
                    "checksum": lambda: brief(bench_checksum(ctx, short=True)),
                    # not a BASELINE configuration: small alphabets, which only the second, reference-like
                    # search of the dense-class pass brings within 3 % of the reference's size
                    "few_values": lambda: brief(bench_deflate(ctx, "fewvalues", short=True))}
            for name, leg in legs.items():
                try:
                    res.setdefault("other_workloads", {})[name] = leg()
                except Exception as e:  # noqa: BLE001  (a secondary leg must not cost the headline line)
                    res.setdefault("other_workloads", {})[name] = {"error": repr(e)}
                torch.cuda.empty_cache()
            try:
                res["end_to_end"] = end_to_end(ctx)
            except Exception as e:  # noqa: BLE001
                res["end_to_end"] = {"error": repr(e)}
    if args.workload == "deflate" and world > 1 and not args.no_extras:
        # The path a Node process takes on a multi-GPU host: ONE process, host buffers, the batch split over
        # the devices named in zgpu_init.  Rank 0 runs it over all the node's GPUs while the other ranks wait
        # (their GPUs are idle by then); reported beside the per-rank figure, never as `value`.
        barrier()
        if rank == 0:
            try:
                res["in_process_host_path"] = host_multi_device(ctx, list(range(min(world, ndev))))
            except Exception as e:  # noqa: BLE001
                res["in_process_host_path"] = {"error": repr(e)}
        barrier()
    if rank == 0:
        print(json.dumps(res), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


def brief(r):
    keep = ("value", "unit", "ms_per_step", "config", "ratio_out_over_in", "roofline", "roundtrip_all_units",
            "units_verified_on_gpu", "units_failed", "ratio_vs_ref", "ratio_vs_ref_by_kind", "cpu_baseline", "crc_checked",
            "bit_exact", "combine_checked")
    return {k: r[k] for k in keep if k in r}


# --------------------------------------------------------------------------
def make_workload(ctx, workload, short):
    """-> (data tensor, [(in_off, ctype, flags, label)], description, config dict)"""
    args, N, dev, rank = ctx["args"], ctx["N"], ctx["dev"], ctx["rank"]
    seed = 1234 + rank
    if workload == "deflate":
        nu, ub = args.units or 65536, args.unit_bytes or 65536
        data = gen_units(nu, ub, seed, dev)
        off = np.arange(nu + 1, dtype=np.uint64) * np.uint64(ub)
        desc = ("batched RawDeflate, %d x %d B independent units of mixed text/binary per GPU, compressionType %d, "
                "lazy %d" % (nu, ub, args.ctype, args.lazy))
        return data, [(off, args.ctype, N.F_FINAL_EACH, "units")], desc, {"units_per_gpu": nu, "unit_bytes": ub}
    if workload == "zip":
        nf, mx = args.units or (32768 if short else 262144), args.unit_bytes or 16384
        data, off = gen_files(nf, mx, seed, dev)
        # files alternate FIXED / DYNAMIC (per-file deflateOption): one engine call per group, as
        # Zip.compress() batches them; the even files are packed first, then the odd ones
        sizes = (off[1:] - off[:-1]).astype(np.int64)
        order = np.concatenate([np.arange(0, nf, 2), np.arange(1, nf, 2)])
        nfx = (nf + 1) // 2
        starts = torch.from_numpy(off[:-1][order].astype(np.int64)).to(dev)
        lens = torch.from_numpy(sizes[order]).to(dev)
        new_off = np.zeros(nf + 1, dtype=np.uint64)
        new_off[1:] = np.cumsum(sizes[order].astype(np.uint64))
        packed = torch.empty(int(new_off[-1]), dtype=torch.uint8, device=dev)
        for lo in range(0, nf, 32768):  # a slice of the files at a time: the index tensors are 8 bytes per byte moved
            hi = min(nf, lo + 32768)
            b0, b1 = int(new_off[lo]), int(new_off[hi])
            seg = torch.repeat_interleave(torch.arange(hi - lo, device=dev), lens[lo:hi])
            first = torch.from_numpy((new_off[lo:hi] - new_off[lo]).astype(np.int64)).to(dev)
            packed[b0:b1] = data[starts[lo:hi][seg] + (torch.arange(b1 - b0, device=dev) - first[seg])]
        data = packed
        desc = ("Zip.compress() batch: %d files of 1-%d B (uniform, mean %.1f KiB) of mixed text/binary per GPU, "
                "half FIXED half DYNAMIC, CRC-32 per file" % (nf, mx, float(new_off[-1]) / nf / 1024))
        return data, [(new_off[: nfx + 1].copy(), 1, N.F_FINAL_EACH | N.F_WANT_CRC, "FIXED files"),
                      (new_off[nfx:].copy(), 2, N.F_FINAL_EACH | N.F_WANT_CRC, "DYNAMIC files")], desc, {
                          "files_per_gpu": nf, "max_file_bytes": mx}
    if workload == "fewvalues":
        # units of two to six equally likely byte values: the content the bounded search cannot serve and the
        # library's dense-class pass is for (every unit is classified on the device and searched by the thorough variant)
        nu, ub = args.units or 512, args.unit_bytes or 65536
        g = torch.Generator(device=dev)
        g.manual_seed(seed ^ 0xFE)
        k = (torch.arange(nu, device=dev) % 5 + 2).unsqueeze(1)
        data = ((torch.rand((nu, ub), device=dev, generator=g) * k).to(torch.uint8) * 37 + 1).to(torch.uint8).view(-1)
        off = np.arange(nu + 1, dtype=np.uint64) * np.uint64(ub)
        desc = ("batched RawDeflate, %d x %d B units of 2..6 equally likely byte values per GPU, compressionType %d, lazy %d"
                % (nu, ub, args.ctype, args.lazy))
        return data, [(off, args.ctype, N.F_FINAL_EACH, "units")], desc, {"units_per_gpu": nu, "unit_bytes": ub}
    if workload == "gzip":
        nc, cb = args.units or (1024 if short else 4096), args.unit_bytes or (1 << 20)
        data = gen_units(nc * (cb // 65536), 65536, seed, dev)
        off = np.arange(nc + 1, dtype=np.uint64) * np.uint64(cb)
        # one gzip member per GiB: its chunks are consecutive units of one chunked, primed stream
        per = max(1, (1 << 30) // cb)
        plans = []
        for lo in range(0, nc, per):
            hi = min(nc, lo + per)
            plans.append((off[lo:hi + 1].copy(), args.ctype,
                          N.F_SYNC_ALIGN | N.F_FINAL_LAST | N.F_PRIME_WINDOW | N.F_WANT_CRC, "member %d" % (lo // per)))
        desc = ("Gzip of a %.0f GiB corpus per GPU in %d B chunks (one member per GiB, chunks primed with the 32 KiB "
                "before them), compressionType %d, CRC-32 per chunk combined on the host" % (nc * cb / 2**30, cb, args.ctype))
        return data, plans, desc, {"chunks_per_gpu": nc, "chunk_bytes": cb}
    raise ValueError(workload)


def bench_deflate(ctx, workload, short=False, ctype=None):
    args, z, N, dev, rank, world, dist = (ctx[k] for k in ("args", "z", "N", "dev", "rank", "world", "dist"))
    if ctype is not None:  # a secondary leg under another compressionType
        args = argparse.Namespace(**dict(vars(args), ctype=ctype))
        ctx = dict(ctx, args=args)
    barrier, red_dev = ctx["barrier"], ctx["red_dev"]
    steps, warmup = (2, 1) if short else (args.steps, args.warmup)
    data, groups, desc, cfg = make_workload(ctx, workload, short)
    torch.cuda.empty_cache()  # the generator's temporaries go back to the device before the library allocates
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    plans, outs = [], []
    for off, ctype, flags, _ in groups:
        p = z.Plan(off, ctype, args.lazy, flags)
        plans.append(p)
        outs.append(torch.empty(p.out_bound + 64, dtype=torch.uint8, device=dev))

    def step():
        for p, o in zip(plans, outs):
            p.run(data.data_ptr(), o.data_ptr(), o.numel(), stream)

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    # the library's HIP events bracket the kernels of the last run of every plan on this stream
    k_ms = search_ms = emit_ms = 0.0
    out_bytes = 0
    fetched = []
    for p, (_, _, flags, _) in zip(plans, groups):
        a, _ = p.kernel_ms()
        s, e = p.phase_ms()
        k_ms, search_ms, emit_ms = k_ms + a, search_ms + s, emit_ms + e
        off, crc, _ = p.fetch(want_crc=bool(flags & N.F_WANT_CRC))
        fetched.append((off, crc))
        out_bytes += int(off[-1])
    in_bytes = int(data.numel())

    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    tot_in = torch.tensor([float(in_bytes)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot_in, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    value = float(tot_in.item()) * steps / dt / 1e9

    cfg = dict(cfg)
    cfg["workload"] = desc
    cfg["parallelism"] = "units sharded over %d GPU(s), no collective" % world
    cfg["search"] = {"max_chain": args.chain or N.DEFAULT_MAX_CHAIN, "nice_length": args.nice,
                     "deep_chain": args.deep if args.deep >= 0 else N.DEFAULT_DEEP_CHAIN}
    res = {
        "metric": METRIC if workload == "deflate" else "deflate input GB/s (%s workload)" % workload,
        "value": round(value, 3), "unit": "GB/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic", "config": cfg,
        "ratio_out_over_in": round(out_bytes / in_bytes, 5),
    }
    if rank == 0:
        # read the input once, write the compressed stream once; the path is two kernels
        # (match search, then Huffman coding + bit emission), priced together
        algo_bytes = in_bytes + out_bytes
        # units of at most 16 KiB run in the search variants with several workgroups per CU and the
        # wave-per-unit emit variant
        small_units = all(int((np.diff(np.asarray(g[0])).max() if len(g[0]) > 1 else 0)) <= 16384 for g in groups)
        traffic, traffic_src = measured_traffic(workload, cfg, args)
        res["roofline"] = {"bound": "hbm", "achieved": round(algo_bytes / (k_ms * 1e-3) / 1e9, 3),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(algo_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                           "traffic": traffic, "traffic_source": traffic_src,
                           "kernel": ("zgpu_deflate_quarter_kernel (units up to 8 KiB) / zgpu_deflate_half_kernel + "
                                      "zgpu_emit_small_kernel" if small_units else
                                      "zgpu_deflate_kernel + zgpu_deflate_thorough_kernel (second search of the "
                                      "classified units; inside search_kernel_ms) + zgpu_emit_kernel"
                                      if workload == "fewvalues" else
                                      "zgpu_deflate_kernel + zgpu_emit_kernel"), "kernel_ms": round(k_ms, 3),
                           "search_kernel_ms": round(search_ms, 3), "emit_kernel_ms": round(emit_ms, 3),
                           "timing": "HIP events recorded by the library on the run's stream around the kernels of "
                                     "the last step (measured in this run)"}
        lds = measured_lds(workload, cfg, args)
        if lds:
            res["search_kernel_occupancy"] = lds  # what actually bounds the path: VALU issue and the LDS pipe
        res.update(verify_all(data, groups, outs, fetched, dev))
        if workload == "gzip":
            res.update(check_stream_crc(ctx, data, groups, fetched))
        if world == 1 and not args.no_cpu:
            res.update(cpu_leg(args, data, groups, outs, fetched, 4.0 if short else args.cpu_seconds,
                               kind_of=unit_kind if workload == "deflate" else None))
    for p in plans:
        p.close()
    return res


def measured_traffic(workload, cfg, args):
    """HBM-side bytes per launch from the committed rocprofv3 counter pass of this exact workload
    (profiles/traffic.json, written by tools/summarize_profiles.py) -- replayed, not measured in
    this run; None for any other workload."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None, None
    w = t.get("workload", {})
    if workload != "deflate" or (w.get("units"), w.get("unit_bytes"), w.get("ctype"), w.get("lazy")) != (
            cfg.get("units_per_gpu"), cfg.get("unit_bytes"), args.ctype, args.lazy):
        return None, None
    return t.get("bytes_per_launch"), ("replayed from profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                       "passes of this workload, %s); not measured in this run" % t.get("source"))


def measured_lds(workload, cfg, args):
    """LDS-pipe and VALU occupancy of the search kernel from the same committed counter passes
    (measured on the same 65536-unit batch), or None for another workload."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return None
    w = t.get("workload", {})
    if workload != "deflate" or (w.get("unit_bytes"), w.get("ctype"), w.get("lazy")) != (
            cfg.get("unit_bytes"), args.ctype, args.lazy):
        return None
    lds = dict(t.get("lds") or {})
    if lds:
        lds["source"] = ("replayed from profiles/traffic.json (rocprofv3 --pmc SQ_* passes on this 65536-unit batch, %s); "
                         "not measured in this run" % t.get("source"))
    return lds


def verify_all(data, groups, outs, fetched, dev):
    """Every unit of the batch through the GPU round-trip verifier (a strict DEFLATE decoder
    that compares with the input as it goes; test infrastructure, outside the timed region)."""
    try:
        import _gpu_verify as gv
        t0 = time.perf_counter()
        bad = total = 0
        for (off, _, flags, _), out, (out_off, _) in zip(groups, outs, fetched):
            chunked = bool(flags & 2) and not (flags & 1)
            st = gv.verify(out, torch.from_numpy(out_off.astype(np.int64)).to(dev), data,
                           torch.from_numpy(off.astype(np.int64)).to(dev), final_each=not chunked,
                           primed=bool(flags & 8))
            bad += int((st != 0).sum())
            total += int(st.numel())
        return {"roundtrip_all_units": bad == 0, "units_verified_on_gpu": total, "units_failed": bad,
                "verify_ms": round((time.perf_counter() - t0) * 1e3, 1)}
    except OSError as e:  # the verifier library is optional for the timing itself
        return {"roundtrip_all_units": None, "verify_error": str(e)}


def check_stream_crc(ctx, data, groups, fetched):
    """gzip workload: every member's CRC-32 by combine of its chunks' values against one pass
    over the member's bytes."""
    N = ctx["N"]
    ok = True
    for (off, _, _, _), (_, crc) in zip(groups, fetched):
        c = 0
        for i in range(off.size - 1):
            c = N.crc32_combine(c, int(crc[i]), int(off[i + 1] - off[i]))
        whole, _ = N.checksum_device(data.data_ptr() + int(off[0]), int(off[-1] - off[0]))
        ok = ok and c == whole
    return {"crc_checked": bool(ok)}


KIND_NAMES = ("text", "incrementing runs", "random", "low entropy")


def unit_kind(u):
    """content kind of unit u of gen_units (index into KIND_NAMES)"""
    k = u % 20
    return 0 if k < 10 else 1 if k < 15 else 2 if k < 18 else 3


def cpu_leg(args, data, groups, outs, fetched, seconds, kind_of=None):
    """Oracle on a bounded sample of the same units: baseline speed (one thread, then every
    core of the box's share over disjoint units), size ratio -- in total and, where the batch's
    layout tells the content kind of a unit (kind_of), per kind -- and round trip."""
    import math
    import _oracle as zo
    zo.lib()
    sample = []
    for (off, ctype, flags, _), out, (out_off, _) in zip(groups, outs, fetched):
        n = off.size - 1
        stride = max(1, n // (2048 // len(groups)))
        while kind_of is not None and stride > 1 and math.gcd(stride, 20) != 1:
            stride += 1  # the kinds repeat every 20 units: a stride that shares a factor with 20 would skip some
        for u in range(0, n, stride):
            sample.append((off, out, out_off, u, ctype, bool(flags & 1)))
    host = {}

    def fetch(i):
        off, out, out_off, u, _, _ = sample[i]
        if i not in host:
            host[i] = (data[int(off[u]):int(off[u + 1])].cpu().numpy(),
                       out[int(out_off[u]):int(out_off[u + 1])].cpu().numpy())
        return host[i]

    # one thread
    t_used, n_done, ref_bytes, gpu_bytes, in_b, ok = 0.0, 0, 0, 0, 0, True
    by_kind = [[0, 0, 0] for _ in KIND_NAMES]  # units, reference bytes, GPU bytes
    for i in range(len(sample)):
        if t_used > seconds * 0.5:
            break
        unit, comp = fetch(i)
        ctype, alone = sample[i][4], sample[i][5]
        t0 = time.perf_counter()
        ref = zo.rawdeflate(unit, ctype, args.lazy)
        t_used += time.perf_counter() - t0
        if alone:
            ok = ok and zo.inflate_raw(comp, unit.size) == unit.tobytes()
        ref_bytes += len(ref)
        gpu_bytes += comp.size
        if kind_of is not None:
            row = by_kind[kind_of(sample[i][3])]
            row[0], row[1], row[2] = row[0] + 1, row[1] + len(ref), row[2] + comp.size
        in_b += unit.size
        n_done += 1
    one = in_b / max(t_used, 1e-9) / 1e9
    # every core: the same units again, one thread per core, each thread one C call over its share
    # of the units per round (the oracle call releases the GIL)
    nth = host_threads()
    by_type = {}
    for i in range(n_done):
        by_type.setdefault(sample[i][4], []).append(fetch(i)[0])
    packs = []  # (flat bytes, offsets, ctype) per thread
    for ctype, us in by_type.items():
        for t in range(nth):
            mine = us[t::nth]
            if mine:
                o = np.zeros(len(mine) + 1, dtype=np.uint64)
                o[1:] = np.cumsum([u.size for u in mine])
                packs.append((np.concatenate(mine), o, ctype))
    reps = max(1, int(min(seconds * 0.5, 10.0) * one * 1e9 * nth / max(in_b, 1)))

    def work(pk):
        for _ in range(reps):
            zo.rawdeflate_many(pk[0], pk[1], pk[2], args.lazy)
        return pk[0].size * reps

    t0 = time.perf_counter()
    with ThreadPoolExecutor(nth) as ex:
        done_bytes = sum(ex.map(work, packs))
    t_all = time.perf_counter() - t0
    all_rate = done_bytes / t_all / 1e9
    extra = {}
    if kind_of is not None:
        # the summed ratio weighs every kind by its compressed bytes (incrementing runs and low-entropy units
        # compress to 1-3 %): each kind against the reference on its own units
        extra["ratio_vs_ref_by_kind"] = {KIND_NAMES[k]: {"ratio": round(g / max(r, 1), 5), "units": c}
                                         for k, (c, r, g) in enumerate(by_kind) if c}
    return {
        **extra,
        "cpu_baseline": {"value": round(all_rate, 5), "unit": "GB/s", "cores": nth, "kind": "port",
                         "single_core_value": round(one, 5), "cpu_model": cpu_model(),
                         "sample": "%d of the same units (evenly spaced over the batch), C restatement of "
                                   "src/rawdeflate.js; one thread for single_core_value, then the same units %d x over "
                                   "%d threads (one per core of this box's share)" % (n_done, reps, nth)},
        "ratio_vs_ref": round(gpu_bytes / max(ref_bytes, 1), 5), "roundtrip_ok": bool(ok), "sample_units": n_done,
    }


def host_batch_rate(N, in_ptr, off, out_ptr, cap, reps=4):
    """best rate of zgpu_deflate_batch over `reps` calls on host buffers (GB/s of input), and the output size"""
    n = off.size - 1
    out_off = np.zeros(n + 1, dtype=np.uint64)
    L = N.lib()
    best = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = L.zgpu_deflate_batch(in_ptr, off.ctypes.data, n, 2, 0, N.F_FINAL_EACH, out_ptr, cap, out_off.ctypes.data,
                                  None, None)
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError("zgpu_deflate_batch rc %d" % rc)
        best = max(best, float(off[-1] - off[0]) / dt / 1e9)
    return best, int(out_off[-1])


def end_to_end(ctx):
    """The drop-in path with PCIe in it: the host-buffer entry point through the C ABI on 1 GiB of the
    default workload -- the caller's buffers in ordinary memory, then in page-locked memory from
    zgpu_host_alloc -- and through the N-API addon under Node when Node is there."""
    N, dev = ctx["N"], ctx["dev"]
    nu, ub = 16384, 65536
    data = gen_units(nu, ub, 1, dev).cpu().numpy()
    off = (np.arange(nu + 1, dtype=np.uint64) * np.uint64(ub))
    cap = int(N.deflate_bound(ub) + 48) * nu + 64
    out = np.empty(cap, dtype=np.uint8)
    best, size = host_batch_rate(N, data.ctypes.data, off, out.ctypes.data, cap)
    res = {"c_abi_host_buffers": {"value": round(best, 2), "unit": "GB/s",
                                  "what": "zgpu_deflate_batch on 1 GiB (16384 x 64 KiB units) in pageable host memory, "
                                          "input upload and output download included, best of 4"}}
    pin_in = pin_out = None
    try:
        pin_in, pin_out = N.HostBuffer(data.size), N.HostBuffer(cap)
        pin_in.array[:] = data
        best, size2 = host_batch_rate(N, pin_in.ptr, off, pin_out.ptr, cap)
        same = size2 == size and bytes(pin_out.array[:size]) == bytes(out[:size])
        res["c_abi_page_locked_buffers"] = {"value": round(best, 2), "unit": "GB/s", "same_bytes": bool(same),
                                            "what": "the same call with input and output in zgpu_host_alloc memory"}
        best, _ = host_batch_rate(N, pin_in.ptr, off, out.ctypes.data, cap)
        res["c_abi_page_locked_input"] = {"value": round(best, 2), "unit": "GB/s",
                                          "what": "input in zgpu_host_alloc memory, output in pageable memory"}
    except Exception as e:  # noqa: BLE001
        res["c_abi_page_locked_buffers"] = {"error": repr(e)}
    finally:
        for b in (pin_in, pin_out):
            if b is not None:
                b.close()
    node = shutil.which("node")
    script = os.path.join(ROOT, "zlib.es_amd", "js", "test", "bench_addon.js")
    if node and os.path.exists(script):
        try:
            r = subprocess.run([node, script, "--json"], capture_output=True, text=True, timeout=120)
            res["node_addon"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            res["node_addon"] = {"error": repr(e)}
    return res


def host_multi_device(ctx, devices):
    """zgpu_deflate_batch in ONE process over the devices named in zgpu_init: 1 GiB of the default workload
    per device, in page-locked host memory.  With one device this is end_to_end's c_abi_page_locked_buffers."""
    N, dev = ctx["N"], ctx["dev"]
    per, ub = 16384, 65536
    nu = per * len(devices)
    off = (np.arange(nu + 1, dtype=np.uint64) * np.uint64(ub))
    cap = int(N.deflate_bound(ub) + 48) * nu + 64
    pin_in, pin_out = N.HostBuffer(nu * ub), N.HostBuffer(cap)
    ids = (ctypes.c_int32 * len(devices))(*devices)
    try:
        for d in range(len(devices)):
            pin_in.array[d * per * ub:(d + 1) * per * ub] = gen_units(per, ub, 1 + d, dev).cpu().numpy()
        rc = N.lib().zgpu_init(ids, len(devices))
        if rc != 0:
            raise RuntimeError("zgpu_init rc %d" % rc)
        best, size = host_batch_rate(N, pin_in.ptr, off, pin_out.ptr, cap)
    finally:
        N.lib().zgpu_init(None, 0)
        pin_in.close()
        pin_out.close()
    return {"value": round(best, 2), "unit": "GB/s", "devices": len(devices), "ratio_out_over_in": round(size / (nu * ub), 5),
            "what": "one process, zgpu_init over %d device(s), zgpu_deflate_batch on %d GiB in page-locked host memory "
                    "(units split by bytes over the devices, each device's host threads on its local CPUs), "
                    "PCIe included, best of 4" % (len(devices), len(devices))}


def bench_checksum(ctx, short=False):
    args, z, dev, rank, world, barrier = (ctx[k] for k in ("args", "z", "dev", "rank", "world", "barrier"))
    nbytes = (65536 * 65536) if short else (args.units or 65536) * (args.unit_bytes or 65536)
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
        n = min(nbytes, (64 << 20) if short else (256 << 20))
        host = data[:n].cpu().numpy()
        t0 = time.perf_counter()
        c, a = zo.crc32(host), zo.adler32(host)
        t = time.perf_counter() - t0
        gc, ga = z._native.checksum_device(data.data_ptr(), n, stream)
        res["cpu_baseline"] = {"value": round(n / t / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                               "cpu_model": cpu_model(),
                               "sample": "first %d bytes, C restatement of src/crc32.js + src/adler32.js" % n}
        res["bit_exact"] = bool((c, a) == (gc, ga))
        # the whole buffer against the combine of its quarters (the reference has no combine: pinned against
        # its resumable update in the tests)
        q = nbytes // 4
        cc, aa = 0, 1
        for i in range(4):
            lo, hi = i * q, (nbytes if i == 3 else (i + 1) * q)
            pc, pa = z._native.checksum_device(data.data_ptr() + lo, hi - lo, stream)
            cc, aa = z._native.crc32_combine(cc, pc, hi - lo), z._native.adler32_combine(aa, pa, hi - lo)
        res["combine_checked"] = bool((cc, aa) == (crc, ad))
    return res


if __name__ == "__main__":
    main()
