#!/bin/bash
# Round 2, GPU call 11: graph replay against eager launches (one graph; two graphs around the RCCL all-reduce), then the
# default bench command (graph replay by default, with its eager / fp32 / default / rccl_1rank side legs and the CPU baseline).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02k; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_graph_gpu.py -q -p no:cacheprovider --timeout 500 2>&1 | tail -15 | tee $O/pytest_graph.log
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch']); print(d['roofline'])
for k in ('fp32_path','eager_path','default_path','rccl_1rank'): print(k, {a: b for a, b in d.get(k, {}).items() if a in ('value','ms_per_step','launch','error')})
print(d.get('cpu_baseline', {}).get('value'))"
grep -i "capture\|error" $O/bench.err | head -5
