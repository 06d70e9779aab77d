#!/bin/bash
# Same-box A/B of one environment switch on the default bench line: scripts/r04_ab_env.sh NAME OFF_VALUE ON_VALUE [reps]
R=${GRAFT_REPO_ROOT:-$PWD}; N=$1; A=$2; B=$3; REPS=${4:-2}; O=$R/gpurun_out/r04ab_$N; mkdir -p $O
for rep in $(seq $REPS); do for v in $A $B; do
  env $N=$v timeout 300 python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-variants 2>$O/${v}_$rep.err | tail -1 > $O/${v}_$rep.json
  python -c "import sys,json; d=json.loads(open('$O/${v}_$rep.json').read()); print('$N=$v', d['value'], d['ms_per_step'], d['final_loss'])"
done; done
