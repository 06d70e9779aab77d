#!/bin/bash
# token weight-gradient kernels: GPU parity test, then twgrad vs the 1x1 case of conv_wgrad per step shape
R=${GRAFT_REPO_ROOT:-$PWD}; T=${TAG:-r05h}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
if [ -z "$SKIP_TESTS" ]; then timeout 420 python -m pytest tests/test_fused_gpu.py -m gpu -x -q -p no:cacheprovider -k "token_wgrad" 2>&1 | tail -4 | tee $O/pytest_twgrad.log; fi
timeout 600 python -m monodetr_amd.tools.wgradbench --out $O/wgradbench.json "$@" > $O/wgradbench.log 2>&1
tail -2 $O/wgradbench.log | cut -c1-300
python - <<PY
import json
d = json.load(open("$O/wgradbench.json"))
for k, r in d.items():
    v = sorted(((kk[:-3], x) for kk, x in r.items() if kk.endswith("_us") and kk != "bound_us" and x), key=lambda t: t[1])
    print("%-26s bound %5.1f | " % (k, r["bound_us"]) + "  ".join("%s %.1f" % (a, b) for a, b in v))
PY
