#!/bin/bash
# Round 4, call 1: the MSDA backward candidates of branch next/msda-prologue on a GPU for the first time -- tests, then the
# operator's kernel times (opbench, bf16-native, initial offsets) for main's build / this tree / LPS=2, alternating; the token GEMM forms.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_msda_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_msda.log 2>&1; echo "pytest default rc=$?"; tail -1 $O/pytest_msda.log
MDETR_MSDA_LPS=2 timeout 400 python -m pytest tests/test_msda_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_msda_lps2.log 2>&1; echo "pytest lps2 rc=$?"; tail -1 $O/pytest_msda_lps2.log
ob() {  # name, env...
    local name=$1; shift
    env "$@" timeout 120 python -m monodetr_amd.tools.opbench --dtype bf16 --dist init --iters 50 > $O/op_$name.json 2>$O/op_$name.err
    python - $O/op_$name.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
e, c = d['encoder'], d['decoder']
print('%-14s enc bwd %.4f ms %s | fwd %.4f | dec bwd %.4f %s' % (sys.argv[2], e['bwd_ms'], e['bwd_kernels_ms'], e['fwd_ms'], c['bwd_ms'], c['bwd_kernels_ms']))
PY
}
for rep in 1 2; do
    ob main_$rep MDETR_LIB_PATH=$R/monodetr_amd/libmonodetr_amd_main.so
    ob cand_$rep MDETR_NOOP=1
    ob lps2_$rep MDETR_MSDA_LPS=2
done
ob lps2_trained MDETR_MSDA_LPS=2
timeout 200 python -m monodetr_amd.tools.tokenbench --iters 50 --out $O/tokenbench.json > $O/tokenbench.log 2>&1; echo "tokenbench rc=$?"; cut -c1-300 $O/tokenbench.log
