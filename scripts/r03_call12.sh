#!/bin/bash
# Round 3, call 12: attention kernels with the one-multiply dropout decision: tests, op timing, step.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03m; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attn_gpu.py tests/test_trainer_gpu.py tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider --timeout 600 > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_subset.log
grep -n "^E  \|^FAILED" $O/pytest_subset.log | cut -c1-300 | head -12
for p in 0.1 0.0; do timeout 300 python -m monodetr_amd.tools.attnbench --dropout $p 2>/dev/null | tail -1 > $O/attnbench_p$p.json; python -c "
import json; d=json.load(open('$O/attnbench_p$p.json')); print('dropout', d['dropout'], {k: (v['hip']['fwd_TFLOPs'], v['hip']['bwd_TFLOPs'], v['hip']['fwd_ms'], v['hip']['bwd_ms']) for k, v in d.items() if isinstance(v, dict)})"; done
timeout 300 python -m monodetr_amd.tools.attnbench --dropout 0.1 --dtype fp32 2>/dev/null | tail -1 > $O/attnbench_fp32_p0.1.json
timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'], d['roofline']['avg_launch_ms'])" || tail -3 $O/bench.err
