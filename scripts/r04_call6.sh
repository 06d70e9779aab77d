#!/bin/bash
# Round 4, call 6: rolling load pipeline of the own-sample groups (variant build) against the tree's kernel
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04f; mkdir -p $O
cd $R
export TMPDIR=/tmp
ob() {  # name, env...
    local name=$1; shift
    env "$@" timeout 120 python -m monodetr_amd.tools.opbench --dtype bf16 --dist ${DIST:-init} --iters 50 > $O/op_$name.json 2>$O/op_$name.err
    python - $O/op_$name.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
e, c = d['encoder'], d['decoder']
print('%-14s enc bwd %.4f ms %s | fwd %.4f | dec bwd %.4f %s' % (sys.argv[2], e['bwd_ms'], e['bwd_kernels_ms'], e['fwd_ms'], c['bwd_ms'], c['bwd_kernels_ms']))
PY
}
ob base_1 MDETR_NOOP=1
for v in $(ls monodetr_amd/variants/ | sed 's/lib_//; s/.so//' | grep -v phases); do
    ob $v MDETR_LIB_PATH=$R/monodetr_amd/variants/lib_$v.so
    MDETR_LIB_PATH=$R/monodetr_amd/variants/lib_$v.so timeout 300 python -m pytest tests/test_msda_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1
done
ob base_2 MDETR_NOOP=1
