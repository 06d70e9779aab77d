#!/bin/bash
# Round 2, GPU call 15: graph test (bars from the measured eager spread); operator / shape breakdown after GroupNorm, Linear
# routing, batch-first depth encoder; MSDA backward at 16 waves with 24 x 32 tiles.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02o; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_graph_gpu.py tests/test_model_gpu.py -x -q -p no:cacheprovider --timeout 500 2>&1 | tail -4 | tee $O/pytest.log
ob() { echo "== $1 ${2:-init} ${3:-bf16}"; env $1 timeout 120 python -m monodetr_amd.tools.opbench --dist ${2:-init} --dtype ${3:-bf16} --iters 30 2>&1 | tail -1 | tee $O/opbench_$(echo "$1$2$3" | tr -c 'A-Za-z0-9\n' '_').json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e, c = d['encoder'], d['decoder']
print('   encoder fwd %.4f bwd %.4f ms %s | decoder bwd %.4f ms %s' % (e['fwd_ms'], e['bwd_ms'], e['bwd_kernels_ms'], c['bwd_ms'], c['bwd_kernels_ms']))"; }
ob "MDETR_MSDA_TILE_H=24 MDETR_MSDA_TILE_W=32"
ob "MDETR_MSDA_TILE_H=16 MDETR_MSDA_TILE_W=32"
ob "MDETR_MSDA_TILE_H=24 MDETR_MSDA_TILE_W=32" trained
timeout 300 python -m monodetr_amd.tools.stepprof --top 120 > $O/stepprof_bf16.txt 2>$O/stepprof.err; head -125 $O/stepprof_bf16.txt | cut -c1-200; tail -2 $O/stepprof.err | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-variants 2>$O/bench.err | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch'][:20], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
