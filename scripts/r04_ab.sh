#!/bin/bash
# developer A/B inside one box: the tree's library and every monodetr_amd/variants/lib_*.so, alternating, REPS rounds (opbench, bf16, init)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${TAG:-ab}; mkdir -p $O
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
if [ -n "$PYTEST" ]; then timeout 400 python -m pytest tests/test_msda_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_msda.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_msda.log; fi
for rep in $(seq 1 ${REPS:-2}); do
    ob tree_$rep MDETR_NOOP=1
    for v in $(ls monodetr_amd/variants/ 2>/dev/null | sed 's/lib_//; s/.so//'); do
        ob ${v}_$rep MDETR_LIB_PATH=$R/monodetr_amd/variants/lib_$v.so
    done
done
