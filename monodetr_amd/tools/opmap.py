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
    xc = torch._C._profiler._ExperimentalConfig(verbose=True) if want_stacks else None       # (Python frames need the verbose collector)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=bool(want_stacks),
                 experimental_config=xc) as prof:
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
        evs = list(prof.events())

        def frames_of(e):
            fr = [f for f in (e.stack or []) if "monodetr_amd" in f or "bench.py" in f][:3]
            return " <- ".join(f.split("monodetr_amd/")[-1] for f in fr)
        # a backward node carries the sequence number of the forward operator that recorded it: its launches go to that operator's frames
        fwd = {}
        for e in evs:
            if getattr(e, "sequence_nr", -1) >= 0 and e.stack and not e.name.startswith("autograd::engine"):
                w = frames_of(e)
                if w:
                    fwd.setdefault(e.sequence_nr, w)
        for e in evs:
            if "all" not in want_stacks and e.name not in want_stacks:
                continue
            dt = sum(k.duration for k in e.kernels) if getattr(e, "kernels", None) else 0.0
            if dt <= 0:
                continue
            where = frames_of(e)
            if not where:
                up, node = e, None
                while up is not None:
                    if up.name.startswith("autograd::engine::evaluate_function"):
                        node = up
                    up = up.cpu_parent
                if node is not None:
                    where = "backward (%s) of: %s" % (node.name.split(": ")[-1], fwd.get(node.sequence_nr, "?"))
                else:
                    where = "(no Python frame)"
            key = (e.name, str(e.input_shapes)[:70], where)
            groups[key][0] += len(e.kernels)
            groups[key][1] += dt
        lines.append("")
        lines.append("%9s %6s  operator / shapes / innermost frames of this repo" % ("us", "calls"))
        for (name, shp, where), (n, dt) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
            lines.append("%9.1f %6d  %s %s  %s" % (dt, n, name, shp, where))
    text = "\n".join(lines)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
