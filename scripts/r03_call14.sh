#!/bin/bash
# Round 3, call 14: output-channel block of conv3x3 at the under-occupied stages (A/B), decimate without the stray copies.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03o; mkdir -p $O
cd $R
export TMPDIR=/tmp
for nb in 4 2 1; do MDETR_CONV3X3_NB=$nb timeout 200 python -m monodetr_amd.tools.convbench --only conv3x3 --iters 30 2>/dev/null | tail -1 > $O/conv3x3_nb$nb.json; python -c "
import json; d=json.load(open('$O/conv3x3_nb$nb.json')); print('nb', $nb, {k: (v['ms'], v['TFLOPs']) for k, v in d.items() if k.startswith('conv3x3')})"; done
timeout 200 python -m monodetr_amd.tools.convbench --only conv3x3 --iters 30 2>/dev/null | tail -1 > $O/conv3x3_auto.json; python -c "
import json; d=json.load(open('$O/conv3x3_auto.json')); print('auto', {k: (v['ms'], v['TFLOPs']) for k, v in d.items() if k.startswith('conv3x3')})"
timeout 300 python -m pytest tests/test_fused_gpu.py -x -q -m gpu -p no:cacheprovider -k "conv3x3 or decimate" 2>&1 | tail -2
b() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'])" || tail -3 $O/bench_$tag.err; }
b auto X=1
b nb4 MDETR_CONV3X3_NB=4
