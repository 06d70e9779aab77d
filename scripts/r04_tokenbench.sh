#!/bin/bash
# Kernel-level times of the token GEMM forms against the library per shape (rocprofv3 kernel trace of tools/tokenbench)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04tg; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/trace_tg
PYTHONPATH=$R timeout 100 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_tg -- python -m monodetr_amd.tools.tokenbench --iters 20 --only "${ONLY:-encoder_256to256,encoder_packed,decoder_256to256,layer1_64to256,depth_tokens}" --out $O/tokenbench.json > $O/tokenbench.log 2>&1
f=$(find /tmp/trace_tg -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "token_gemm" in n or n.startswith("Cijk"):
        k = (n[:110], r["Grid_Size_X"])
        a = agg.setdefault(k, [0, 0])
        a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for (n, g), (c, t) in agg.items():
    print("%5d x %8.2f us  grid %-9s %s" % (c, t / c / 1e3, g, n))
PY
grep -o '"forms_bit_equal": [a-z]*\|"max_err_vs_library": [0-9.e-]*' $O/tokenbench.json | sort | uniq -c
