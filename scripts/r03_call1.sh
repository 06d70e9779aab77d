#!/bin/bash
# Round 3, call 1: the product's replayed training iteration (tests/test_trainer_gpu.py, the entry-point test), the 512x1760 /
# 100-query model tests, the refactored bench (config2 / config5 side objects, clocks).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_trainer_gpu.py tests/test_graph_gpu.py "tests/test_model_gpu.py" "tests/test_fused_gpu.py::test_train_val_entry_point_end_to_end" -q -p no:cacheprovider --timeout 900 -s > $O/pytest_new.log 2>&1; echo "pytest rc=$?"
grep -n "passed\|failed\|error\|PG-CHILD\|worst\|spread\|launch mode" $O/pytest_new.log | tail -30
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch'], d['config'].get('gpu_clocks')); print(d['roofline'])
for k in ('fp32_path','eager_path','default_path','rccl_1rank','config2','config5'): print(k, {a: b for a, b in d.get(k, {}).items() if a in ('value','ms_per_step','launch','error')})
print({k: v for k, v in d.get('cpu_baseline', {}).items() if k in ('value','cores','kind','s_per_iter','note')})"
tail -5 $O/bench.err
