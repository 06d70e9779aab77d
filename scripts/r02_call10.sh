#!/bin/bash
# Round 2, GPU call 10: operator / shape breakdown of the step; eager bench twice (run-to-run spread of the main leg);
# whole-step hipGraph replay with the committed bf16 list.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02j; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m monodetr_amd.tools.stepprof --top 110 > $O/stepprof_bf16.txt 2>$O/stepprof.err; head -115 $O/stepprof_bf16.txt | cut -c1-200; tail -2 $O/stepprof.err | cut -c1-200
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-variants 2>$O/bench_eager$i.err | tail -1 > $O/bench_eager$i.json
python -c "
import json; d=json.load(open('$O/bench_eager$i.json')); print('eager', {k: d[k] for k in ('value','ms_per_step','final_loss')}, d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
timeout 400 python bench.py --graph on --no-cpu-baseline --no-variants 2>$O/bench_graph.err | tail -1 > $O/bench_graph.json
python -c "
import json; d=json.load(open('$O/bench_graph.json')); print('graph', {k: d[k] for k in ('value','ms_per_step','final_loss')}, d['roofline']['frac'])" || tail -5 $O/bench_graph.err
