"""Experiment: the captured iteration's hipGraph as a DOT file (node kinds, edges) -- are there forks, joins or non-kernel
nodes where the replayed step idles (profiles/r04gap_*.txt)?  Writes gpurun_out/r04dot/graph.dot."""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import monodetr_amd  # noqa: F401,E402  (runtime flags before torch's HIP runtime starts)
import torch  # noqa: E402
import bench  # noqa: E402

_Graph = torch.cuda.CUDAGraph


class DebugGraph(_Graph):
    def __new__(cls, *a, **k):
        return super().__new__(cls, True)                  # keep_graph: the hipGraph_t stays reachable (raw_cuda_graph)

    def __init__(self, *a, **k):
        super().__init__(True)


torch.cuda.CUDAGraph = DebugGraph
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
chosen, _ = bench.committed_switches("bf16")
step = bench.TrainStep(dev, 8, "bf16", graph=True, switches=chosen)
step.strict = True
print(step.try_capture(), flush=True)
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "r04dot")
os.makedirs(out, exist_ok=True)
import ctypes  # noqa: E402
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGraphDebugDotPrint.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint]
rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(step.graph.raw_cuda_graph()), os.path.join(out, "graph.dot").encode(), 0)
print("hipGraphDebugDotPrint ->", rc, flush=True)
print("dumped", os.path.getsize(os.path.join(out, "graph.dot")), "bytes", flush=True)
