#!/bin/bash
# Step-level A/B in one GPU call: the committed bf16 list against variants of it.
#   VARIANTS="tgemm:MDETR_TGEMM=1;other:MDETR_X=1,MDETR_Y=0"  REPS=2 STEPS=60 bash scripts/r05_step_ab.sh [tag]
# (a family variable in the environment replaces the committed list, so every run spells the list out)
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r05ab}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
LIST=$(python -c "import bench; print(' '.join(k + '=1' for k in sorted(bench.COMMITTED_SWITCHES['bf16'])))" 2>/dev/null)
run() { # name, extra env...
  local name=$1; shift
  env $LIST "$@" timeout 400 python bench.py --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline --no-variants 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "import json; d=json.loads(open('$O/$name.json').read()); print('$name', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss', d.get('final_loss'), d['config'].get('launch'))" 2>/dev/null || { echo "$name FAILED"; tail -5 $O/$name.err; }
}
if [ -n "$PRETEST" ]; then timeout 600 python -m pytest $PRETEST -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5 | tee $O/pretest.log; fi
for rep in $(seq 1 ${REPS:-2}); do
  run committed_$rep
  IFS=';' read -ra VS <<< "${VARIANTS:-tgemm:MDETR_TGEMM=1}"
  for v in "${VS[@]}"; do
    name=${v%%:*}; envs=${v#*:}
    run ${name}_$rep ${envs//,/ }
  done
done
