"""profiles/msda_pmc_traffic.json from a `tools/pmc_summary` file of the MSDA backward's counter passes: HBM bytes per call of the
bf16-native operator at the encoder shape = sum over its kernels (pre-pass, one-pass kernel, finalize; the row with the largest
grid of each) of 2 x FETCH_SIZE + WRITE_SIZE KiB (the guide's gfx950 correction for 16-byte lane loads), stamped with the hash of
the kernel sources (`bench.kernel_source_sha`) so that `bench.py` reports it only for the tree it was measured on.

    python -m monodetr_amd.tools.pmc_traffic_record <pmc_summary.json> [--out profiles/msda_pmc_traffic.json] [--source NAME]
"""
import argparse
import json
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary")
    ap.add_argument("--out", default="")
    ap.add_argument("--source", default="")
    a = ap.parse_args()
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
    sys.path.insert(0, root)
    import bench
    rows = json.load(open(a.summary))
    best = {}
    for r in rows:
        name = r["kernel"]
        key = "absmax" if "absmax" in name else "fused" if "bwd_fused" in name else "finalize" if "finalize" in name else None
        if key is None or "FETCH_SIZE" not in r or "WRITE_SIZE" not in r:
            continue
        if key not in best or r["grid"] > best[key]["grid"]:
            best[key] = r
    if "fused" not in best:
        raise SystemExit("no msda_bwd_fused row with FETCH_SIZE / WRITE_SIZE in %s" % a.summary)
    total = sum(int((2 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024) for r in best.values())
    out = a.out or os.path.join(root, "profiles", "msda_pmc_traffic.json")
    rec = json.load(open(out)) if os.path.exists(out) else {}
    rec["msda_backward_bf16_Lq10200"] = total
    rec["kernel_source_sha"] = bench.kernel_source_sha()
    rec["source"] = a.source or os.path.basename(a.summary)
    rec["rows"] = {k: {"kernel": r["kernel"][:60], "grid": r["grid"], "FETCH_SIZE_KiB": round(r["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(r["WRITE_SIZE"], 1)}
                   for k, r in best.items()}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({"msda_backward_bf16_Lq10200": total, "kernel_source_sha": rec["kernel_source_sha"], "rows": rec["rows"]}))


if __name__ == "__main__":
    main()
