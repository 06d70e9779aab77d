#!/bin/bash
# Round 3, call 23: the optimizer's gradient gather: tests, step A/B.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03g2; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_trainer_gpu.py tests/test_graph_gpu.py -x -q -m gpu -p no:cacheprovider --timeout 600 -k "adamw or trainer or replay or graph or process_group or e2e" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_subset.log
grep -n "^E  \|^FAILED" $O/pytest_subset.log | cut -c1-300 | head -8
b() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'])" || tail -3 $O/bench_$tag.err; }
b gather X=1
b foreach MDETR_ADAMW_GATHER=0
b gather_eager MDETR_BENCH_GRAPH=off
b foreach_eager MDETR_BENCH_GRAPH=off MDETR_ADAMW_GATHER=0
