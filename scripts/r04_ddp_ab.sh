#!/bin/bash
# One-rank RCCL form of the step, same box: the gradient gather recorded in the graphs (1) against issued by the host (0),
# with the plain single-process step beside them
R=${GRAFT_REPO_ROOT:-$PWD}; T=${TAG:-r04ddpab}; O=$R/gpurun_out/$T; mkdir -p $O
run() { timeout 300 python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-variants 2>$O/$1.err | tail -1 > $O/$1.json
  python -c "import sys,json; d=json.loads(open('$O/$1.json').read()); print('$1', d['value'], d['ms_per_step'], d['final_loss'], d['config']['launch'][:40])"; }
for rep in 1 2; do
  run plain_$rep
  MDETR_BENCH_FORCE_DDP=1 MDETR_GATHER_IN_GRAPH=1 run ddp_in_graph_$rep
  MDETR_BENCH_FORCE_DDP=1 MDETR_GATHER_IN_GRAPH=0 run ddp_host_cat_$rep
  MDETR_BENCH_FORCE_DDP=1 MDETR_BENCH_SYNC=flat run ddp_flat_$rep
done
