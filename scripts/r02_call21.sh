#!/bin/bash
# Round 2, GPU call 21: after restricting the batched split-K to > 8 192 rows (default path accuracy) and giving the column sums
# more workgroups on few-row inputs: the model / colsum / fused tests, then the bench line.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02u; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_colsum_gpu.py tests/test_fused_gpu.py -q -p no:cacheprovider --timeout 600 > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_subset.log | tail -2
timeout 400 python bench.py --no-cpu-baseline --no-variants 2>$O/bench.err | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch'][:20], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
