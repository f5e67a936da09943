"""zgpu_plan_run inside a HIP graph: the call allocates nothing and waits for nothing, and the stream the plan
owns (the first search runs on it beside the search of the dense units) forks from the capturing stream by an
event and joins it by another, so the whole call is captured -- and every replay writes the same bytes as the
host-buffer call."""
import numpy as np
import pytest

import _oracle as zo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ctype", [1, 2])
def test_plan_run_is_captured_and_replayed(ctype):
    import torch
    assert torch.cuda.is_available()
    import zlib_es_amd as z
    dev = torch.device("cuda")
    datas = [zo.gen({"gen": "text", "n": 60000, "seed": 1}), zo.gen({"gen": "small-alpha", "n": 65536, "seed": 2, "k": 3}),
             zo.gen({"gen": "rnd", "n": 30000, "seed": 3}), zo.gen({"gen": "runs", "n": 65536, "seed": 3}),
             zo.gen({"gen": "small-alpha", "n": 20000, "seed": 5, "k": 2})]
    datas = [np.ascontiguousarray(d) for d in datas]
    want, _, _ = z.deflate_batch(datas, ctype, 0)
    off = np.zeros(len(datas) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([d.size for d in datas])
    flat = torch.from_numpy(np.concatenate(datas)).to(dev)
    plan = z.Plan(off, ctype, 0, 1)
    out = torch.empty(plan.out_bound + 64, dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        plan.run(flat.data_ptr(), out.data_ptr(), out.numel(), torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        out_off, _, _ = plan.fetch()
        host = out.cpu().numpy()
        for i, w in enumerate(want):
            assert bytes(host[int(out_off[i]):int(out_off[i + 1])]) == bytes(w), (rep, i)
    del graph
    plan.close()
