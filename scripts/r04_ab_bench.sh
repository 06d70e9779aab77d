#!/bin/bash
# same-box A/B of the whole step: "$@" = env assignments of variant B (variant A = the tree as it is); two rounds, alternating
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${TAG:-abb}; mkdir -p $O
cd $R; export TMPDIR=/tmp
one() { local name=$1; shift
  env "$@" timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', d['value'], d['ms_per_step'], d['final_loss'])"; }
for rep in 1 2; do one A_$rep MDETR_NOOP=1; one B_$rep "$@"; done
