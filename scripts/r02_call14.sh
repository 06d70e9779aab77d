#!/bin/bash
# Round 2, GPU call 14: graph capture BEFORE the process group exists (test + bench with all side legs); MSDA backward launch
# variants (12-wave workgroups); GroupNorm in the committed list.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02n; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_graph_gpu.py -x -q -s -p no:cacheprovider --timeout 500 > $O/pytest_graph.log 2>&1; echo "graph test rc=$?"; grep -n "^spread\|passed\|failed\|Error" $O/pytest_graph.log | head -20 | cut -c1-300
ob() { echo "== $1 ${2:-init} ${3:-bf16}"; env $1 timeout 120 python -m monodetr_amd.tools.opbench --dist ${2:-init} --dtype ${3:-bf16} --iters 30 2>&1 | tail -1 | tee $O/opbench_$(echo "$1$2$3" | tr -c 'A-Za-z0-9\n' '_').json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e, c = d['encoder'], d['decoder']
print('   encoder fwd %.4f bwd %.4f ms %s | decoder bwd %.4f ms %s' % (e['fwd_ms'], e['bwd_ms'], e['bwd_kernels_ms'], c['bwd_ms'], c['bwd_kernels_ms']))"; }
ob "MDETR_MSDA_THREADS=1024"
ob "MDETR_MSDA_THREADS=768 MDETR_MSDA_GROUPS=2"
ob "MDETR_MSDA_THREADS=768 MDETR_MSDA_GROUPS=4"
ob "MDETR_MSDA_THREADS=768 MDETR_MSDA_GROUPS=4 MDETR_MSDA_TILE_H=24 MDETR_MSDA_TILE_W=32"
ob "MDETR_MSDA_THREADS=768 MDETR_MSDA_GROUPS=4" trained
timeout 500 python bench.py --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch'][:20], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
for k in ('fp32_path','eager_path','default_path','rccl_1rank'): print('   ', k, {a: b for a, b in d.get(k, {}).items() if a in ('value','ms_per_step','error','launch')})" || tail -5 $O/bench.err
