"""Workgroup rounds per launch from a rocprofv3 --kernel-trace CSV.

    python -m monodetr_amd.tools.rounds <kernel_trace.csv> [--steps K] [--top 40]

A launch whose workgroups do not all fit on the device at once runs in ROUNDS: the dispatcher refills a CU when its workgroups
end, and a last, partly filled round costs a whole workgroup duration (csrc/conv3x3.hip: 576 workgroups on 512 places took 13 us
for the first 512 and 17 us for the last 64).  For every (kernel, grid) of the last K steps: workgroups, places = 256 CUs x
workgroups per CU (LDS bytes, registers, 32 waves), rounds = workgroups / places, and the time a perfect last round would save,
(ceil(rounds) - rounds) / ceil(rounds) of the launch -- an upper bound, summed per step.
"""
import argparse
import csv
import math
from collections import defaultdict


def per_cu(wg_threads, lds, vgpr, agpr):
    waves = max(1, (wg_threads + 63) // 64)
    regs = max(1, vgpr + agpr)
    regs = (regs + 7) // 8 * 8
    by_regs = (512 // regs) * 4 // waves if regs <= 512 else 0            # waves per SIMD x 4 SIMDs
    by_waves = 32 // waves
    by_lds = (160 * 1024) // lds if lds > 0 else 10 ** 6
    return max(1, min(by_regs, by_waves, by_lds))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=0, help="keep the kernels of the last K steps (a step = 3 encoder msda_bwd_fused launches)")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps = 1
    if a.steps > 0:
        marks = [i for i, r in enumerate(rows) if "msda_bwd_fused" in r["Kernel_Name"] and int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]) > 100000]
        if len(marks) >= 3 * a.steps + 1:
            rows = rows[marks[-(3 * a.steps + 1)] + 1:]
            steps = a.steps
    agg = defaultdict(lambda: [0, 0.0, None])
    for r in rows:
        g = lambda *ks: next((int(float(r[k])) for k in ks if k in r and r[k] not in ("", None)), 0)        # noqa: E731
        grid = g("Grid_Size_X", "Grid_Size") * max(1, g("Grid_Size_Y")) * max(1, g("Grid_Size_Z"))
        wg = g("Workgroup_Size_X", "Workgroup_Size") * max(1, g("Workgroup_Size_Y")) * max(1, g("Workgroup_Size_Z"))
        lds, vgpr, agpr = g("LDS_Block_Size", "LDS_Block_Size_v"), g("VGPR_Count"), g("Accum_VGPR_Count")
        key = (r["Kernel_Name"].replace("mdetr::(anonymous namespace)::", "").replace("void ", "")[:70], grid, wg, lds, vgpr, agpr)
        e = agg[key]
        e[0] += 1
        e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out = []
    for (name, grid, wg, lds, vgpr, agpr), (n, us, _) in agg.items():
        wgs = grid // max(1, wg)
        cap = per_cu(wg, lds, vgpr, agpr)
        places = 256 * cap
        rounds = wgs / places
        waste = (math.ceil(rounds) - rounds) / math.ceil(rounds) if rounds > 0 else 0.0
        out.append((us * waste / steps, us / steps, n / steps, name, wgs, cap, rounds, waste, lds, vgpr + agpr))
    out.sort(reverse=True)
    print("%9s %9s %6s %7s %4s %7s %6s %6s %5s  kernel" % ("tail_us", "us/step", "calls", "wgs", "/CU", "rounds", "waste", "lds", "regs"))
    for t, us, n, name, wgs, cap, rounds, waste, lds, regs in out[:a.top]:
        print("%9.1f %9.1f %6.1f %7d %4d %7.2f %6.2f %6d %5d  %s" % (t, us, n, wgs, cap, rounds, waste, lds, regs, name))
    print("sum of tails: %.1f us/step of %.1f us/step" % (sum(o[0] for o in out), sum(o[1] for o in out)))


if __name__ == "__main__":
    main()
