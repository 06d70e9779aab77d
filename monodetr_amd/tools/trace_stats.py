"""Steady-state per-kernel statistics from a rocprofv3 --kernel-trace CSV.

    python -m monodetr_amd.tools.trace_stats <kernel_trace.csv> --steps K [--out stats.csv] [--top 40]

rocprofv3's own --stats covers the whole process, i.e. also MIOpen's solver search during the first
warm-up steps (its naive reference convolutions then dominate the table).  This tool keeps only the
kernels of the LAST K training steps: a step is delimited by the encoder-shaped msda_bwd_fused (or msda_bwd_d32) launches
(3 per step), so the window starts after the (3*K+1)-th last of them ended.
"""
import argparse
import csv
import json
from collections import defaultdict


def category(name):
    """Coarse owner of a kernel, for the per-category table of DESIGN.md 6."""
    low = name.lower()
    if "igemm" in low or "ck::" in name or "subtensorop" in low or "miopen" in low or "naive_conv" in low:
        return "MIOpen convolutions"
    if name.startswith("Cijk"):
        return "hipBLASLt GEMMs"
    if "mdetr" in name:
        for key, cat in (("msda", "MSDA (this repo)"), ("attn", "dense attention (this repo)"), ("colsum", "column sums (this repo)"),
                         ("lsa", "matching (this repo)"), ("pair_losses", "fused losses (this repo)"), ("ddn", "fused losses (this repo)"),
                         ("adamw", "fused AdamW (this repo)")):
            if key in name:
                return cat
        return "other kernels of this repo"
    if "multi_tensor" in name:
        return "multi-tensor (optimizer, folds)"
    if "layer_norm" in low or "gradgammabeta" in low or "gradinput" in low or "group_norm" in low or "rowwisemoments" in low:
        return "normalisation layers"
    if "reduce_kernel" in name:
        return "framework reductions"
    if "elementwise" in name or "vectorized" in name:
        return "framework elementwise"
    if "rocclr" in name:
        return "runtime copies / fills"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--out")
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--skip-last", type=int, default=0, help="leave out the last N steps of the trace (bench.py's side measurements "
                    "after the timed region launch differently)")
    ap.add_argument("--copies", help="rocprofv3 --memory-copy-trace CSV of the same run: copy-engine transfers join the timeline "
                    "as MEMCPY <direction> entries (they are not kernels, and otherwise look like idle time)")
    ap.add_argument("--context", type=int, default=0, help="with --gaps: the launches around the first occurrence of the N largest groups")
    ap.add_argument("--gaps", type=float, default=0.0, help="also list the idle intervals longer than this many microseconds "
                    "(no kernel of any stream running), grouped by the kernels either side")
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0), r.get("Queue_Id", "?"), int(r.get("Scratch_Size") or 0)))
    if a.copies:
        with open(a.copies) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY %s %s bytes" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?"))),
                             0, "copy", 0))
    rows.sort()
    bwd = [r for r in rows if "msda_bwd_fused" in r[2]] or [r for r in rows if "msda_bwd_d32" in r[2]]
    big = max(r[3] for r in bwd)
    enc = [r for r in bwd if r[3] == big]
    need, skip = 3 * a.steps, 3 * a.skip_last
    assert len(enc) > need + skip, "trace holds %d encoder backward launches, need > %d" % (len(enc), need + skip)
    t_start = enc[-need - skip - 1][1]       # end of the last kernel of the step before the window
    t_end = enc[-skip - 1][1] if skip else rows[-1][1]
    win = [r for r in rows if t_start <= r[0] < t_end]
    agg = defaultdict(lambda: [0, 0, 10 ** 18, 0])
    for s, e, name, *_ in win:
        x = agg[name]
        x[0] += 1; x[1] += e - s; x[2] = min(x[2], e - s); x[3] = max(x[3], e - s)
    total = sum(v[1] for v in agg.values())
    table = sorted(agg.items(), key=lambda kv: -kv[1][1])
    wall = (t_end - t_start) / 1e6
    print(json.dumps({"window_ms": round(wall, 2), "ms_per_step_wall": round(wall / a.steps, 2),
                      "gpu_busy_ms_per_step": round(total / 1e6 / a.steps, 2), "kernels_per_step": round(len(win) / a.steps, 1)}))
    if a.gaps > 0:
        # (per interval: the kernels either side, the hardware queues they ran on -- a change means a cross-queue signal --
        # and the scratch bytes per work-item of the kernel that follows -- a scratch user may wait for its allocation)
        idle, pairs, covered, last, lastq = 0, defaultdict(lambda: [0, 0, -1]), win[0][1], win[0][2], win[0][4]
        for i, (s_, e_, name, _, q, scratch) in enumerate(win[1:], 1):
            if s_ > covered:
                idle += s_ - covered
                if s_ - covered >= a.gaps * 1e3:
                    x = pairs[(last[:60], "%s  [queue %s%s%s]" % (name[:60], lastq, "" if q == lastq else " -> " + q,
                                                                 ", scratch %d" % scratch if scratch else ""))]
                    x[0] += 1; x[1] += s_ - covered
                    if x[2] < 0:
                        x[2] = i
            if e_ > covered:
                covered, last, lastq = e_, name, q
        print("idle (no kernel running) %.3f ms/step; intervals >= %.0f us:" % (idle / 1e6 / a.steps, a.gaps))
        ranked = sorted(pairs.items(), key=lambda kv: -kv[1][1])
        for (before, after), (n, t, _) in ranked[:30]:
            print("  %7.1f us/step %5.1f x/step avg %6.1f us   %s  ->  %s" % (t / 1e3 / a.steps, n / a.steps, t / n / 1e3, before, after))
        for (before, after), (n, t, at) in ranked[:a.context]:
            print("  -- first occurrence of [%s -> %s], times in us relative to the interval's end:" % (before[:40], after[:40]))
            for s_, e_, name, grid, q, scratch in win[max(at - 5, 0):at + 4]:
                print("     %9.1f .. %9.1f  q%s grid %-8d %s" % ((s_ - win[at][0]) / 1e3, (e_ - win[at][0]) / 1e3, q, grid, name[:100]))
    if a.out:
        with open(a.out, "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "CallsPerStep", "TotalMsPerStep", "AverageUs", "Percentage", "MinUs", "MaxUs"])
            for name, (n, t, mn, mx) in table:
                w.writerow([name[:160], round(n / a.steps, 2), round(t / 1e6 / a.steps, 4), round(t / n / 1e3, 2),
                            round(100.0 * t / total, 2), round(mn / 1e3, 2), round(mx / 1e3, 2)])
    cats = defaultdict(lambda: [0, 0])
    for name, (n, t, mn, mx) in table:
        c = cats[category(name)]
        c[0] += n; c[1] += t
    for cat, (n, t) in sorted(cats.items(), key=lambda kv: -kv[1][1]):
        print("%-36s %8.2f ms/step %7.0f launches/step" % (cat, t / 1e6 / a.steps, n / a.steps))
    for name, (n, t, mn, mx) in table[:a.top]:
        print("%6.2f%% %9.3f ms/step %7.1f calls/step avg %9.1f us  %s" % (100.0 * t / total, t / 1e6 / a.steps, n / a.steps, t / n / 1e3, name[:110]))


if __name__ == "__main__":
    main()
