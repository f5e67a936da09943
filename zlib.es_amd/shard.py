"""Sharding of independent units over the GPUs of one node (one process per
GPU, torch.distributed).  There is no data-path collective: every rank
compresses a contiguous, byte-balanced range of units on its own GPU and the
host concatenates the results; the only cross-shard arithmetic is the left
fold of CRC-32 / Adler-32 combine over per-chunk values.
"""
import numpy as np

from . import _native as N


def partition_by_bytes(sizes, world):
    """Contiguous unit ranges [(lo, hi)] per rank, split where the running byte
    count crosses k/world of the total (every unit lands in exactly one range)."""
    sizes = np.asarray(sizes, dtype=np.int64)
    n = sizes.size
    csum = np.concatenate([[0], np.cumsum(sizes)])
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        i = int(np.searchsorted(csum, target, side="left"))
        cuts.append(min(max(i, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def combine_checksums(crcs, adlers, lens):
    """Whole-stream (crc32, adler32) from per-chunk values, in order."""
    crc, adler = 0, 1
    for c, a, l in zip(crcs, adlers, lens):
        crc = N.crc32_combine(crc, c, l)
        adler = N.adler32_combine(adler, a, l)
    return crc, adler


def _gpu_compress(units, ctype, lazy, flags, history=None):
    return N.deflate_batch(units, ctype, lazy, flags, want_crc=True, want_adler=True, history=history)


def compress_sharded(units, ctype=N.DYNAMIC, lazy=0, flags=N.F_FINAL_EACH, compress_fn=None):
    """Compress `units` (same list on every rank) across the ranks of the default
    process group.  Rank 0 returns (outputs, crc32s, adler32s) for all units in
    order; other ranks return None.  Without an initialised process group the
    call runs on the local GPU alone.  `compress_fn(units, ctype, lazy, flags, history)`
    replaces the GPU call (tests use it to exercise the sharding on CPU); history is the
    tail of the units before the rank's range when the units are chunks of one primed stream."""
    fn = compress_fn or _gpu_compress
    try:
        import torch.distributed as dist
        active = dist.is_available() and dist.is_initialized()
    except ImportError:
        active = False
    if not active:
        return fn(units, ctype, lazy, flags, None)
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = partition_by_bytes([len(u) for u in units], world)[rank]
    # FINAL_LAST belongs to the globally last unit only
    local_flags = flags
    if (flags & N.F_FINAL_LAST) and hi != len(units):
        local_flags &= ~N.F_FINAL_LAST
    # chunks of one stream keep their window across the ranks' boundaries: the bytes before a
    # rank's first chunk are on every rank, so they go along as history
    history = None
    if (flags & N.F_PRIME_WINDOW) and (flags & N.F_SYNC_ALIGN) and lo > 0 and hi > lo:
        tail, need = [], 32768
        for u in reversed(units[:lo]):
            a = N._u8(u)
            tail.append(a[-need:] if need < a.size else a)
            need -= tail[-1].size
            if need <= 0:
                break
        history = np.concatenate(tail[::-1]) if tail else None
    mine = fn(units[lo:hi], ctype, lazy, local_flags, history) if hi > lo else ([], [], [])
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(mine, gathered, dst=0)
    if rank != 0:
        return None
    outs, crcs, adlers = [], [], []
    for part in gathered:
        outs += list(part[0])
        crcs += list(part[1] or [])
        adlers += list(part[2] or [])
    return outs, crcs, adlers


def gzip_chunked(data, chunk_size=1 << 20, ctype=N.DYNAMIC, lazy=0, mtime=0, compress_fn=None,
                 independent=False):
    """One gzip member whose body is made of chunks compressed side by side
    (each byte-aligned by an empty stored block, BFINAL on the last), with the
    CRC-32 trailer obtained by combine.  A chunk may match into the 32 KiB before
    it (across the ranks' boundaries as well) unless `independent`.
    Container framing as src/gzip.js:67-83, :117-126.  Returns bytes on rank 0
    (None on other ranks)."""
    a = N._u8(data)
    chunks = [a[i:i + chunk_size] for i in range(0, a.size, chunk_size)] or [a]
    flags = N.F_SYNC_ALIGN | N.F_FINAL_LAST | (0 if independent else N.F_PRIME_WINDOW)
    res = compress_sharded(chunks, ctype, lazy, flags, compress_fn)
    if res is None:
        return None
    outs, crcs, adlers = res
    crc, _ = combine_checksums(crcs, adlers, [c.size for c in chunks])
    head = bytes([0x1F, 0x8B, 8, 0, mtime & 0xFF, (mtime >> 8) & 0xFF, (mtime >> 16) & 0xFF, (mtime >> 24) & 0xFF,
                  0, 255])
    il = a.size & 0xFFFFFFFF
    tail = bytes([crc & 0xFF, (crc >> 8) & 0xFF, (crc >> 16) & 0xFF, (crc >> 24) & 0xFF,
                  il & 0xFF, (il >> 8) & 0xFF, (il >> 16) & 0xFF, (il >> 24) & 0xFF])
    return head + b"".join(bytes(o) for o in outs) + tail


def zlib_chunked(data, chunk_size=1 << 20, ctype=N.DYNAMIC, lazy=0, compress_fn=None, independent=False):
    """One zlib stream (RFC 1950: header as src/deflate.js:49-82, body, big-endian Adler-32 as
    :94-97) whose body is made of chunks compressed side by side; the Adler-32 of the whole
    input comes from the chunks' values by combine.  Returns bytes on rank 0 (None elsewhere)."""
    a = N._u8(data)
    chunks = [a[i:i + chunk_size] for i in range(0, a.size, chunk_size)] or [a]
    flags = N.F_SYNC_ALIGN | N.F_FINAL_LAST | (0 if independent else N.F_PRIME_WINDOW)
    res = compress_sharded(chunks, ctype, lazy, flags, compress_fn)
    if res is None:
        return None
    outs, crcs, adlers = res
    _, adler = combine_checksums(crcs, adlers, [c.size for c in chunks])
    cmf = (7 << 4) | 8
    flg = ctype << 6
    flg |= 31 - ((cmf * 256 + flg) % 31)
    tail = bytes([(adler >> 24) & 0xFF, (adler >> 16) & 0xFF, (adler >> 8) & 0xFF, adler & 0xFF])
    return bytes([cmf, flg]) + b"".join(bytes(o) for o in outs) + tail
