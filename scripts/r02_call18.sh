#!/bin/bash
# Round 2, GPU call 18: the other bench configurations under the default (graph replay) launch mode, the N > 1 code path of
# bench.main with one rank (MDETR_BENCH_FORCE_DDP=1), smoke(), MSDA finalize / pre-pass changes.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02r; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 300 python -m pytest tests/test_msda_gpu.py -q -x -p no:cacheprovider --timeout 300 > $O/pytest_msda.log 2>&1; grep -n "passed\|failed" $O/pytest_msda.log
ob() { echo "== $1 ${2:-init} ${3:-bf16}"; env $1 timeout 120 python -m monodetr_amd.tools.opbench --dist ${2:-init} --dtype ${3:-bf16} --iters 30 2>&1 | tail -1 | tee $O/opbench_$(echo "$1$2$3" | tr -c 'A-Za-z0-9\n' '_').json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e, c = d['encoder'], d['decoder']
print('   encoder fwd %.4f bwd %.4f ms %s | decoder bwd %.4f ms %s' % (e['fwd_ms'], e['bwd_ms'], e['bwd_kernels_ms'], c['bwd_ms'], c['bwd_kernels_ms']))"; }
ob "X=1"
ob "X=1" trained
b() { timeout 500 env $1 python bench.py --no-cpu-baseline --no-variants $2 2>$O/bench_$3.err | tail -1 > $O/bench_$3.json; python -c "
import json; d=json.load(open('$O/bench_$3.json')); print('$3', {k: d[k] for k in ('value','ms_per_step','final_loss','n_gpus','dtype')}, d['config']['launch'][:60], d.get('roofline', {}).get('frac'))" || tail -5 $O/bench_$3.err; }
b "X=1" "--config 2" config2
b "X=1" "--config 5" config5
b "X=1" "--precision fp32" fp32
b "MDETR_BENCH_FORCE_DDP=1" "" force_ddp
b "MDETR_BENCH_FORCE_DDP=1" "--graph off" force_ddp_eager
