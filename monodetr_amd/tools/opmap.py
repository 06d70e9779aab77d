"""Which operator of the training iteration launches which kernels: one EAGER iteration of bench.py's step under torch.profiler,
grouped by (operator, input shapes), with the device time and launch count of each group -- the map from the framework's
elementwise / copy / fill / reduce launches in a rocprofv3 trace back to the lines of the model that issue them.

    python -m monodetr_amd.tools.opmap [--top 120] [--out gpurun_out/opmap.txt] [--match add,copy,fill]
"""
import argparse
import os
import sys

import monodetr_amd._runtime_env  # noqa: F401  (before torch)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=150)
    ap.add_argument("--out", default="")
    ap.add_argument("--match", default="", help="comma-separated substrings of operator names to keep (default: all)")
    ap.add_argument("--stacks", default="", help="comma-separated operator names whose launches are also grouped by the repo's Python frames")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
    import bench
    from monodetr_amd.kernel_families import COMMITTED_SWITCHES
    dev = torch.device("cuda", 0)
    step = bench.TrainStep(dev, a.batch, a.precision, switches=COMMITTED_SWITCHES[a.precision], graph=False)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    want_stacks = [m for m in a.stacks.split(",") if m]
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=bool(want_stacks)) as prof:
        step()
        torch.cuda.synchronize()
    keep = [m for m in a.match.split(",") if m]
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        dt = getattr(e, "self_device_time_total", 0) or 0
        if dt <= 0:
            continue
        if keep and not any(m in e.key for m in keep):
            continue
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:150]))
    rows.sort(reverse=True)
    lines = ["%9s %6s  %-46s %s" % ("self_us", "calls", "operator", "input shapes")]
    for dt, n, key, shp in rows[:a.top]:
        lines.append("%9.1f %6d  %-46s %s" % (dt, n, key[:46], shp))
    lines.append("total self device time of the listed groups: %.1f us, %d launches-or-more" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
    if want_stacks:
        import collections
        groups = collections.defaultdict(lambda: [0, 0.0])
        for e in prof.events():
            if e.name not in want_stacks:
                continue
            dt = sum(k.duration for k in e.kernels) if getattr(e, "kernels", None) else 0.0
            if dt <= 0:
                continue
            frames = [f for f in (e.stack or []) if "monodetr_amd" in f or "bench.py" in f][:3]
            where = " <- ".join(f.split("monodetr_amd/")[-1] for f in frames) or "(autograd engine: no Python frame)"
            key = (e.name, str(e.input_shapes)[:70], where)
            groups[key][0] += 1
            groups[key][1] += dt
        lines.append("")
        lines.append("%9s %6s  operator / shapes / innermost frames of this repo" % ("us", "calls"))
        for (name, shp, where), (n, dt) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
            lines.append("%9.1f %6d  %s %s  %s" % (dt, n, name, shp, where))
    text = "\n".join(lines)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
