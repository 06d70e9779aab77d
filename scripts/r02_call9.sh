#!/bin/bash
# Round 2, GPU call 9: MSDA backward with 4 lanes per sample (8 channels each, packed-bf16 dot products, padded LDS cells)
# against the 8-lane layout; fused prediction heads; bf16 column sums; operator / shape breakdown of the step.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02i; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_msda_gpu.py tests/test_colsum_gpu.py tests/test_model_gpu.py -q -p no:cacheprovider --timeout 300 2>&1 | tail -3 | tee $O/pytest_subset.log
ob() { echo "== $1 ${2:-init} ${3:-bf16}"; env $1 timeout 120 python -m monodetr_amd.tools.opbench --dist ${2:-init} --dtype ${3:-bf16} --iters 30 2>&1 | tail -1 | tee $O/opbench_$(echo "$1$2$3" | tr -c 'A-Za-z0-9\n' '_').json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e, c = d['encoder'], d['decoder']
print('   encoder fwd %.4f bwd %.4f ms %s | decoder bwd %.4f ms %s' % (e['fwd_ms'], e['bwd_ms'], e['bwd_kernels_ms'], c['bwd_ms'], c['bwd_kernels_ms']))"; }
ob "MDETR_MSDA_LPS=4"
ob "MDETR_MSDA_LPS=8"
ob "MDETR_MSDA_LPS=4" trained
ob "MDETR_MSDA_LPS=8" trained
ob "MDETR_MSDA_LPS=4" init fp32
timeout 400 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','fp32_path','default_path','rccl_1rank','roofline') if k in d}); print(d.get('cpu_baseline')); print(d.get('launches_per_step'))"
timeout 300 python -m monodetr_amd.tools.stepprof --top 90 > $O/stepprof_bf16.txt 2>$O/stepprof.err; head -95 $O/stepprof_bf16.txt | cut -c1-200
