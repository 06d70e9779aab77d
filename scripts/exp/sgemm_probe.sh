#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; D=/tmp/sp_$$_$RANDOM; export TMPDIR=/tmp; cd /tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -- python $R/scripts/exp/sgemm_probe.py | grep CASE > $D.cases
python - $(find $D -name "*kernel_trace.csv" | head -1) $D.cases <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "sgemm" in r["Kernel_Name"]]
cases = [l.split(" ", 1)[1].strip() for l in open(sys.argv[2])]
per = len(rows) // len(cases)
for i, c in enumerate(cases):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0 for r in rows[i * per:(i + 1) * per]]
    print("%-40s grid %-8s %s us" % (c, rows[i * per].get("Grid_Size_X", rows[i * per].get("Grid_Size")), " ".join("%.1f" % x for x in d)))
PY
rm -rf $D $D.cases
