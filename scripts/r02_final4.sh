#!/bin/bash
# Round 2, the driver's round-end commands on the final tree, for the record: pytest -m gpu, smoke(), bench.py.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02y; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider --timeout 600 > $O/pytest_gpu_all.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" $O/pytest_gpu_all.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch']); print(d['roofline'])
for k in ('fp32_path','eager_path','default_path','rccl_1rank'): print(k, {a: b for a, b in d.get(k, {}).items() if a in ('value','ms_per_step','launch','error')})
print({k: v for k, v in d.get('cpu_baseline', {}).items() if k in ('value','cores','kind')})"
