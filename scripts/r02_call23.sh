#!/bin/bash
# Round 2, GPU call 23: weight gradients on a second stream (MDETR_WGRAD_OVERLAP): race tests (bit-identical gradients), then the
# bench with and without it under graph replay and eagerly.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02w; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_fused_gpu.py -x -q -p no:cacheprovider --timeout 500 -k "wgrad_overlap" > $O/pytest_overlap.log 2>&1; echo "rc=$?"; grep -n "passed\|failed\|Error" $O/pytest_overlap.log | tail -4 | cut -c1-300
ALL="MDETR_FUSED_LOSSES=1 MDETR_FUSED_ADAMW=1 MDETR_FUSED_LN=1 MDETR_MSDA_PROLOGUE=1 MDETR_MSDA_BF16=1 MDETR_FUSED_EPILOGUE=1 MDETR_GEMM_RELU=1 MDETR_CONV3X3=1 MDETR_GROUP_NORM=1 MDETR_SMALL_WGRAD=1"
b() { timeout 400 env $1 python bench.py --no-cpu-baseline --no-variants $2 2>$O/bench_$3.err | tail -1 > $O/bench_$3.json; python -c "
import json; d=json.load(open('$O/bench_$3.json')); print('$3', {k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch'][:30])" || tail -5 $O/bench_$3.err; }
b "X=1" "" committed
b "$ALL MDETR_WGRAD_OVERLAP=1" "" overlap
b "$ALL MDETR_WGRAD_OVERLAP=1" "--graph off" overlap_eager
