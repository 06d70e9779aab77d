#!/bin/bash
# Round 4, call 5: block order (tiles first), steps from an LDS counter, count check merged into the read-out, one barrier in the prologue
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04e; mkdir -p $O
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
ob order0 MDETR_MSDA_ORDER=0
ob main_1 MDETR_LIB_PATH=$R/monodetr_amd/libmonodetr_amd_main.so
ob new_2 MDETR_NOOP=1
ob chunks8 MDETR_MSDA_CHUNKS=8
ob chunks16 MDETR_MSDA_CHUNKS=16
ob chunks20 MDETR_MSDA_CHUNKS=20
ob tile16x32 MDETR_MSDA_TILE_H=16
DIST=trained ob new_trained MDETR_NOOP=1
MDETR_LIB_PATH=$R/monodetr_amd/variants/lib_phases.so timeout 120 python -m monodetr_amd.tools.opbench --dtype bf16 --dist init --iters 20 --phases 2>&1 | tee $O/phases.log | grep "^encoder" | cut -c1-600
