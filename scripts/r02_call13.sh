#!/bin/bash
# Round 2, GPU call 13: GroupNorm kernel (tests, then the step with it); graph-vs-eager test with its eager yardstick; the flat
# gradient exchange without the copy back; bench with and without MDETR_GROUP_NORM.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02m; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_fused_gpu.py -x -q -p no:cacheprovider --timeout 500 -k "group_norm" 2>&1 | tail -6 | tee $O/pytest_gn.log
timeout 600 python -X faulthandler -m pytest tests/test_graph_gpu.py -x -q -s -p no:cacheprovider --timeout 500 > $O/pytest_graph.log 2>&1; echo "graph test rc=$?"; grep -n "^eager\|^graph\|^spread\|passed\|failed\|Error" $O/pytest_graph.log | head -20 | cut -c1-300
timeout 300 python -m monodetr_amd.tools.syncbench > $O/syncbench.txt 2>$O/syncbench.err; head -12 $O/syncbench.txt | cut -c1-200
b() { timeout 400 env $1 python bench.py --no-cpu-baseline $2 2>$O/bench_$3.err | tail -1 > $O/bench_$3.json; python -c "
import json; d=json.load(open('$O/bench_$3.json')); print('$3', {k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch'][:20], d['config']['switches'][-3:])
for k in ('fp32_path','eager_path','rccl_1rank'): print('   ', k, {a: b for a, b in d.get(k, {}).items() if a in ('value','ms_per_step','error')})"; }
ALL="MDETR_FUSED_LOSSES=1 MDETR_FUSED_ADAMW=1 MDETR_FUSED_LN=1 MDETR_MSDA_PROLOGUE=1 MDETR_MSDA_BF16=1 MDETR_FUSED_EPILOGUE=1 MDETR_GEMM_RELU=1 MDETR_CONV3X3=1"
b "X=1" "" committed
b "$ALL MDETR_GROUP_NORM=1" "--no-variants" with_gn
