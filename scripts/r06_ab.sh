#!/bin/bash
# Round 6, step-level A/B in one GPU call: optional pre-tests, then the committed bf16 list against variants of it.
#   PRETEST="tests/test_x.py ..." VARIANTS="noheads:-MDETR_HEADS;x:+MDETR_X" REPS=2 STEPS=60 bash scripts/r06_ab.sh [tag]
# A variant is a comma-separated list of +FAMILY / -FAMILY edits of the committed list (or NAME=value for other environment).
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r06ab}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
if [ -n "$PRETEST" ]; then timeout ${PRETEST_TIMEOUT:-900} python -m pytest $PRETEST -m gpu -x -q -p no:cacheprovider -s > $O/pretest.log 2>&1; tail -4 $O/pretest.log; grep -n "^E  \|^FAILED" $O/pretest.log | cut -c1-300 | head -12; fi
run() { # name, edits
  local name=$1 edits=$2
  local envs=$(python - "$edits" <<'PY'
import sys, bench
names = set(bench.COMMITTED_SWITCHES["bf16"]); extra = []
for e in [x for x in sys.argv[1].split(",") if x]:
    if e[0] == "+": names.add(e[1:])
    elif e[0] == "-": names.discard(e[1:])
    else: extra.append(e)
print(" ".join([k + "=1" for k in sorted(names)] + extra))
PY
)
  env $envs timeout 400 python bench.py --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline --no-variants 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "import json; d=json.loads(open('$O/$name.json').read()); print('$name', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss', d.get('final_loss'), 'kernels', d.get('kernels_per_step'))" 2>/dev/null || { echo "$name FAILED"; tail -5 $O/$name.err; }
}
for rep in $(seq 1 ${REPS:-2}); do
  run committed_$rep ""
  IFS=';' read -ra VS <<< "${VARIANTS:-}"
  for v in "${VS[@]}"; do
    [ -z "$v" ] && continue
    run ${v%%:*}_$rep "${v#*:}"
  done
done
python - $O <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/committed_1.json")):
    d = json.load(open(f))
    for r in d.get("families", []): print(r["name"], r["launches"], r["ms_per_step"], r.get("frac"))
PY
