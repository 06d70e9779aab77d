#!/bin/bash
# Per-launch durations of the grouped fp32 head kernels (csrc/sgemm.hip) at the training shape, from a rocprofv3 kernel trace of the
# GPU test of one decoder level (tests/test_sgemm_gpu.py -k heads_level): forward L1 / L2 / L3, backward B1 / B2 / B3, weight gradients.
R=${GRAFT_REPO_ROOT:-$PWD}; D=/tmp/tr_$$_$RANDOM; export TMPDIR=/tmp; cd /tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python -m pytest $R/tests/test_sgemm_gpu.py -q -m gpu -p no:cacheprovider -k heads_level 2>&1 | grep -E "passed|failed" | tail -1
python - $(find $D -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sgemm" in r["Kernel_Name"]]
for r in rows[-7:]:
    print(r["Kernel_Name"].split("(")[0][-28:], "grid", r.get("Grid_Size_X", r.get("Grid_Size")), "wg", r.get("Workgroup_Size_X", r.get("Workgroup_Size")), "%.2f us" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0))
PY
rm -rf $D
