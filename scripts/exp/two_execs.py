"""Experiment: does the host block in hipGraphLaunch until the PREVIOUS launch of the same executable graph has finished?
Two captures of the same iteration, replayed A A A ... against A B A B ...: per-call host time and ms per iteration."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import monodetr_amd  # noqa: F401  (runtime flags before torch's HIP runtime starts)
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
chosen, _ = bench.committed_switches("bf16")
step = bench.TrainStep(dev, 8, "bf16", graph=True, switches=chosen)
print(step.try_capture(), flush=True)
gA = step.graph
step.capture()
gB = step.graph


def run(seq, n=40):
    host = []
    with torch.cuda.stream(step.stream):
        for i in range(10):
            seq[i % len(seq)].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            t = time.perf_counter()
            seq[i % len(seq)].replay()
            host.append(time.perf_counter() - t)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    host.sort()
    return "ms/iteration %.3f   host per launch: median %.3f ms, max %.3f ms; loop returned %.2f ms before the GPU finished" % (
        (t2 - t0) / n * 1e3, host[len(host) // 2] * 1e3, host[-1] * 1e3, (t2 - t1) * 1e3)


for name, seq in (("A A A A", [gA]), ("A B A B", [gA, gB]), ("A A A A", [gA]), ("A B A B", [gA, gB])):
    print(name, run(seq), flush=True)
