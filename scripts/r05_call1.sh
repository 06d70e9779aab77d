#!/bin/bash
# Round 5, call 1: tgemm parity on the GPU, then every token-wise product of the step: library vs tgemm (all tiles) vs the
# weight-in-registers form
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
timeout 420 python -m pytest tests/test_tgemm_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8 | tee $O/pytest_tgemm.log
timeout 600 python -m monodetr_amd.tools.gemmbench --sweep --out $O/gemmbench.json > $O/gemmbench.log 2>&1
tail -3 $O/gemmbench.log | cut -c1-400
python - <<PY
import json
d = json.load(open("$O/gemmbench.json"))
for k, r in d.items():
    tiles = {kk[6:-3]: v for kk, v in r.items() if kk.startswith("tgemm_1") or kk.startswith("tgemm_6")}
    bt = min(tiles, key=tiles.get) if tiles else "-"
    print("%-28s lib %7.1f  tgemm %7.1f  best %7.1f (%s)  regs %s  bound %6.1f" % (k, r["library_us"], r["tgemm_us"], r["tgemm_best_us"], bt, r.get("regs_us", "-"), r["bound_us"]))
PY
