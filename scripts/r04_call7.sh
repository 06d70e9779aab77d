#!/bin/bash
# Round 4, call 7: window rows padded by one cell (vertical neighbours off the same LDS bank)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_msda_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_msda.log 2>&1; echo "pytest default rc=$?"; tail -1 $O/pytest_msda.log
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
ob new_1 MDETR_NOOP=1
ob tile24x31 MDETR_MSDA_TILE_W=31
ob new_2 MDETR_NOOP=1
DIST=trained ob new_trained MDETR_NOOP=1
for v in $(ls monodetr_amd/variants/ 2>/dev/null | sed 's/lib_//; s/.so//'); do
    ob $v MDETR_LIB_PATH=$R/monodetr_amd/variants/lib_$v.so
done
