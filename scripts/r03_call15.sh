#!/bin/bash
# Round 3, call 15: channel-block sweep of conv_taps, workgroup target of conv_wgrad, conv3x3 with the new rule; step.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03p; mkdir -p $O
cd $R
export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', {k.replace('_kernel',''): v['ms'] for k, v in d.items() if k.endswith('_kernel') and ('$3' in k)})"; }
timeout 200 python -m monodetr_amd.tools.convbench --only conv3x3 --iters 30 2>/dev/null | tail -1 > $O/conv3x3_rule.json; show $O/conv3x3_rule.json rule conv3x3
for nb in 1 2 4; do MDETR_CONV_TAPS_NB=$nb timeout 300 python -m monodetr_amd.tools.convbench --only strided --iters 20 2>/dev/null | tail -1 > $O/strided_nb$nb.json; show $O/strided_nb$nb.json taps_nb$nb fwd_; show $O/strided_nb$nb.json taps_nb$nb dgrad_; done
for w in 256 512 1024; do MDETR_CONV_WGRAD_WGS=$w timeout 300 python -m monodetr_amd.tools.convbench --only wgrad --iters 20 2>/dev/null | tail -1 > $O/wgrad_wgs$w.json; show $O/wgrad_wgs$w.json wgs$w wgrad; done
b() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'])" || tail -3 $O/bench_$tag.err; }
b rule X=1
