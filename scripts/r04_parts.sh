#!/bin/bash
# The single-process iteration as ONE executable graph against TWO (forward + criterion | backward + optimizer), same box:
# untraced bench twice each, then the idle intervals of the two-part form
R=${GRAFT_REPO_ROOT:-$PWD}; T=${TAG:-r04parts}; O=$R/gpurun_out/$T; mkdir -p $O
for rep in 1 2; do for parts in 1 2; do
  MDETR_GRAPH_PARTS=$parts timeout 300 python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_p${parts}_$rep.err | tail -1 > $O/bench_p${parts}_$rep.json
  python -c "import sys,json; d=json.loads(open('$O/bench_p${parts}_$rep.json').read()); print('parts=$parts', d['value'], d['ms_per_step'], d['final_loss'], d['config']['launch'][:60])"
done; done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/trace_p2
MDETR_GRAPH_PARTS=2 PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_p2 -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_traced_p2.json 2>$O/bench_traced_p2.err
f=$(find /tmp/trace_p2 -name "*kernel_trace.csv" | head -1)
(cd $R; python -m monodetr_amd.tools.trace_stats $f --steps 8 --skip-last 14 --gaps 8 --context 4 --top 12 > $O/${T}_p2.txt 2>&1); head -14 $O/${T}_p2.txt | cut -c1-200
