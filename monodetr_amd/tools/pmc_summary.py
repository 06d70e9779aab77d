"""Per-kernel averages of rocprofv3 PMC passes (counter_collection CSVs) for this repo's kernels.

    python -m monodetr_amd.tools.pmc_summary <dir-with-rocprofv3-output> [...more dirs] [--match mdetr] [--out x.json]

Each rocprofv3 pass holds a few counters (SQ: 8 slots, TCC: 4, GRBM: 2 -- MI355X_MICROARCH.md); run
one pass per counter set and give all output directories here.  Values are averaged per dispatch over
the LAST `--last` dispatches of each (kernel, grid) -- the first launches of a process include cold
caches.  Also reports VGPR / AGPR / SGPR / LDS as recorded by the profiler.
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def short(name):
    n = name.replace("mdetr::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--match", default="mdetr")
    ap.add_argument("--last", type=int, default=3)
    ap.add_argument("--out")
    a = ap.parse_args()
    vals = defaultdict(lambda: defaultdict(list))          # (kernel, grid) -> counter -> [(dispatch, value)]
    meta = {}
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if a.match not in r["Kernel_Name"]:
                        continue
                    key = (short(r["Kernel_Name"]), int(r.get("Grid_Size") or 0))
                    vals[key][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                    meta[key] = {k: int(float(r[c])) for k, c in (("vgpr", "VGPR_Count"), ("agpr", "Accum_VGPR_Count"),
                                                                 ("sgpr", "SGPR_Count"), ("lds_bytes", "LDS_Block_Size"),
                                                                 ("workgroup", "Workgroup_Size")) if r.get(c) not in (None, "")}
    rows = []
    for key in sorted(vals):
        row = {"kernel": key[0], "grid": key[1], **meta.get(key, {})}
        for cname, lst in sorted(vals[key].items()):
            # a dispatch may appear once per XCD / SE instance: sum instances of one dispatch, then average dispatches
            per = defaultdict(float)
            for disp, v in lst:
                per[disp] += v
            last = [per[k] for k in sorted(per)[-a.last:]]
            row[cname] = sum(last) / len(last)
        c = row
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            row["L2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
            for k, nm in (("SQ_WAIT_ANY", "frac_wave_parked"), ("SQ_WAIT_INST_ANY", "frac_issue_stall"),
                          ("SQ_ACTIVE_INST_ANY", "frac_issuing")):
                if k in c:
                    row[nm] = round(c[k] / c["SQ_WAVE_CYCLES"], 4)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE", 0) > 0:
            # MFMA utilisation = cycles the matrix pipes were busy, summed over the 1 024 SIMDs (256 CUs x 4), over SIMDs x the
            # kernel's duration in cycles.  GRBM_GUI_ACTIVE comes summed over the 8 XCDs (0.49 ms of MSDA backward reads 9.4 M
            # at 2.4 GHz), so the kernel ran GRBM_GUI_ACTIVE / 8 cycles: busy / (1024 x GRBM / 8).  (Rounds 1-3 printed
            # busy / SQ_BUSY_CYCLES -- a per-SE counter under a per-SIMD sum, values 1.5-11 -- which is not a utilisation.)
            row["mfma_utilisation"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * c["GRBM_GUI_ACTIVE"]), 4)
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            row["lds_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        rows.append(row)
    text = json.dumps(rows, indent=1)
    if a.out:
        open(a.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
