"""Developer tool: GPU time of one training step per ATen operator AND input shape (torch.profiler), to find where the
framework's own elementwise / reduction / copy launches of the step come from.

    python -m monodetr_amd.tools.stepprof [--precision bf16] [--top 70] [--steps 4]

The rocprofv3 kernel trace (monodetr_amd/tools/trace_stats.py) says which KERNELS cost time; this says which operator
calls, with which shapes, launched them.  Runs the same step object as bench.py with its committed switch list.
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--min-us", type=float, default=0.0)
    a = ap.parse_args()
    import bench
    from torch.profiler import ProfilerActivity, profile
    # the profiler's shape recorder converts every integer argument to int64: keep the 64-bit dropout seeds below 2^63 here
    from monodetr_amd import add_ln_ext
    host_seed = add_ln_ext._host_seed
    add_ln_ext._host_seed = lambda: host_seed() & (2 ** 63 - 1)
    step = bench.TrainStep(torch.device("cuda", 0), 8, a.precision, switches=bench.committed_switches(a.precision)[0])
    for _ in range(5):
        step.eager_iteration()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(a.steps):
            step.eager_iteration()
        torch.cuda.synchronize()
    rows = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.key_averages(group_by_input_shape=True):
        t = getattr(e, "self_device_time_total", None)
        if t is None:
            t = e.self_cuda_time_total
        if t <= 0:
            continue
        shapes = str([s for s in e.input_shapes if s])[:110]
        r = rows[(e.key, shapes)]
        r[0] += e.count
        r[1] += t
    total = sum(r[1] for r in rows.values())
    print("GPU time per step %.3f ms over %d (operator, shapes) groups" % (total / a.steps / 1e3, len(rows)))
    print("%9s %7s %8s  %s" % ("us/step", "calls", "us/call", "operator  shapes"))
    for (key, shapes), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:a.top]:
        if t / a.steps < a.min_us:
            break
        print("%9.1f %7.1f %8.1f  %s  %s" % (t / a.steps, n / a.steps, t / n, key[:60], shapes))


if __name__ == "__main__":
    main()
