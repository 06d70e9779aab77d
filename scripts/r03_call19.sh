#!/bin/bash
# Round 3, call 19: row softmax for the depth expectation, max-pool kernel: tests + step.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03u; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_model_gpu.py tests/test_backbone_parity_gpu.py -x -q -m gpu -p no:cacheprovider --timeout 600 > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_subset.log
grep -n "^E  \|^FAILED" $O/pytest_subset.log | cut -c1-300 | head -12
b() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'])" || tail -3 $O/bench_$tag.err; }
b new X=1
b new2 X=1
