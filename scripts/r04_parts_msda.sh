#!/bin/bash
# MDETR_GRAPH_PARTS=msda (the second graph starts at the encoder's last MSDA backward launch) against the single graph, same box:
# the two GPU tests of the cut, untraced bench twice each, then the idle intervals of the msda form
R=${GRAFT_REPO_ROOT:-$PWD}; T=${TAG:-r04msda}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_trainer_gpu.py -m gpu -x -q -k "cut_at_the_encoders or (not_the_captured_one and msda)" -p no:cacheprovider 2>&1 | tail -4
bash scripts/r04_ab_env.sh MDETR_GRAPH_PARTS 1 msda 2 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/trace_msda
MDETR_GRAPH_PARTS=msda PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_msda -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_traced.json 2>$O/bench_traced.err
f=$(find /tmp/trace_msda -name "*kernel_trace.csv" | head -1)
(cd $R; python -m monodetr_amd.tools.trace_stats $f --steps 8 --skip-last 14 --gaps 8 --context 3 --top 8 > $O/${T}_gaps.txt 2>&1); head -12 $O/${T}_gaps.txt | cut -c1-200
