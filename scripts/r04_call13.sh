#!/bin/bash
# Round 4, call 13: the head-blocked bf16 forward (msda_fwd_hm.hip) on a GPU: tests, operator times rec / hm, chunk sweep
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04m; mkdir -p $O
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
print('%-14s enc fwd %.4f ms (%.0f GB/s) bwd %.4f | dec fwd %.4f bwd %.4f' % (sys.argv[2], e['fwd_ms'], e['fwd_GBs'], e['bwd_ms'], c['fwd_ms'], c['bwd_ms']))
PY
}
ob rec_1 MDETR_MSDA_FWD=rec
ob hm_1 MDETR_MSDA_FWD=hm
ob rec_2 MDETR_MSDA_FWD=rec
ob hm_2 MDETR_MSDA_FWD=hm
ob hm_ch8 MDETR_MSDA_FWD=hm MDETR_MSDA_FWD_CHUNKS=8
ob hm_ch16 MDETR_MSDA_FWD=hm MDETR_MSDA_FWD_CHUNKS=16
ob hm_ch64 MDETR_MSDA_FWD=hm MDETR_MSDA_FWD_CHUNKS=64
DIST=trained ob hm_trained MDETR_MSDA_FWD=hm
DIST=trained ob rec_trained MDETR_MSDA_FWD=rec
DIST=uniform ob hm_uniform MDETR_MSDA_FWD=hm
DIST=uniform ob rec_uniform MDETR_MSDA_FWD=rec
