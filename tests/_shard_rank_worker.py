"""One rank of tests/test_gpu_sharding_ranks.py: joins a gloo group, runs the sharded calls with the
real GPU call (all ranks share the box's one GPU) and, on rank 0, writes the results to a file.
usage: _shard_rank_worker.py <rank> <world> <port> <result file>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def inputs():
    import _oracle as zo
    units = [zo.gen({"gen": ("text", "rndseq", "runs", "rnd")[i % 4], "n": 20000 + 3001 * i, "seed": 40 + i}) for i in range(13)]
    stream = zo.gen({"gen": "text", "n": 700_000, "seed": 77})
    return units, stream


def main():
    rank, world, port, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)  # before anything touches the GPU
    import zlib_es_amd as z
    try:
        units, stream = inputs()
        res = z.compress_sharded(units, 2, 0)
        gz = z.gzip_chunked(stream, chunk_size=100_000)
        zs = z.zlib_chunked(stream, chunk_size=100_000, ctype=1)
        if rank == 0:
            outs, crcs, adlers = res
            with open(path, "w") as f:
                json.dump({"outs": [bytes(o).hex() for o in outs], "crcs": [int(c) for c in crcs],
                           "adlers": [int(a) for a in adlers], "gz": bytes(gz).hex(), "zs": bytes(zs).hex()}, f)
        else:
            assert res is None and gz is None and zs is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
