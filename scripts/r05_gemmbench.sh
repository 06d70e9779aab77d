#!/bin/bash
# tgemm parity on the GPU, then the step's products: library vs tgemm (tiles / variants).  ARGS: extra gemmbench arguments
R=${GRAFT_REPO_ROOT:-$PWD}; T=${TAG:-r05c}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
if [ -z "$SKIP_TESTS" ]; then timeout 420 python -m pytest tests/test_tgemm_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest_tgemm.log; fi
timeout 900 python -m monodetr_amd.tools.gemmbench --out $O/gemmbench.json "$@" > $O/gemmbench.log 2>&1
tail -2 $O/gemmbench.log | cut -c1-300
python - <<PY
import json
d = json.load(open("$O/gemmbench.json"))
for k, r in d.items():
    v = sorted(((kk[5:-3] or "default", x) for kk, x in r.items() if kk.startswith("tgemm") and kk.endswith("_us") and kk != "tgemm_best_us"), key=lambda t: t[1])
    print("%-28s lib %6.1f regs %5s bound %5.1f | " % (k, r["library_us"], r.get("regs_us", "-"), r["bound_us"]) + "  ".join("%s %.1f" % (a.strip("_[]"), b) for a, b in v[:6]))
PY
