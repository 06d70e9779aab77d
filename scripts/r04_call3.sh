#!/bin/bash
# Round 4, call 3: timing ablations of the MSDA backward (results wrong by construction: where does the time go)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04c; mkdir -p $O
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
ob base MDETR_NOOP=1
for v in $(ls monodetr_amd/variants/ | sed 's/lib_//; s/.so//'); do
    ob $v MDETR_LIB_PATH=$R/monodetr_amd/variants/lib_$v.so
done
ob base2 MDETR_NOOP=1
