"""The N > 1 path with the real GPU call: two gloo ranks (their own processes, started before
anything touches the GPU in them) shard units and the chunks of one stream over "their" GPUs --
the box has one, which both use -- and rank 0 reassembles.  The bytes must be those of one
process doing it all: units are independent, and a chunk's window is primed with the same
32 KiB whether the chunk before it was compressed here or on another rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

import _oracle as zo
import _shard_rank_worker as W

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_with_the_gpu_call(tmp_path):
    import zlib_es_amd as z
    port, path = str(_free_port()), str(tmp_path / "rank0.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_shard_rank_worker.py"), str(r), "2", port, path],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        out, _ = p.communicate(timeout=240)
        logs.append(out.decode(errors="replace")[-2000:])
    assert all(p.returncode == 0 for p in procs), logs
    got = json.load(open(path))
    units, stream = W.inputs()
    outs, crcs, adlers = z.compress_sharded(units, 2, 0)  # no process group here: this process alone
    assert [bytes(o).hex() for o in outs] == got["outs"]
    assert [int(c) for c in crcs] == got["crcs"] == [zo.crc32(u) for u in units]
    assert [int(a) for a in adlers] == got["adlers"]
    gz, zs = bytes.fromhex(got["gz"]), bytes.fromhex(got["zs"])
    assert gz == bytes(z.gzip_chunked(stream, chunk_size=100_000))
    assert zs == bytes(z.zlib_chunked(stream, chunk_size=100_000, ctype=1))
    assert zo.gunzip(gz, stream.size) == stream.tobytes() and zo.inflate_zlib(zs, stream.size) == stream.tobytes()
