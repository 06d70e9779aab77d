#!/bin/bash
# Round 3, call 26: MSDA pre-pass with its maxima spread over eight words: MSDA tests, the operator's kernel times inside the step.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03m2; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_msda_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_msda.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_msda.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['final_loss'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
for k in d['kernels']:
    if 'msda' in k['kernel'] and k.get('Lq') == 10200: print(k['kernel'], k['avg_ms'])"
