#!/bin/bash
# Round 2, GPU call 12: the graph-vs-eager test with its full log; cost of the flat gradient exchange (1-rank RCCL); the model
# tests and a short bench after routing the decoder's / depth encoder's linear layers through token_linear and the level
# embedding's gradient through colsum.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02l; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_graph_gpu.py -x -q -p no:cacheprovider --timeout 500 > $O/pytest_graph.log 2>&1; echo "graph test rc=$?"; grep -n "passed\|failed\|Fatal\|Error" $O/pytest_graph.log | head -8
timeout 300 python -m monodetr_amd.tools.syncbench > $O/syncbench.txt 2>$O/syncbench.err; head -30 $O/syncbench.txt | cut -c1-220
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_fused_gpu.py -x -q -p no:cacheprovider --timeout 500 2>&1 | tail -4 | tee $O/pytest_model.log
timeout 400 python bench.py --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch']); print(d['roofline']['frac'], d['roofline']['avg_launch_ms'])
for k in ('fp32_path','eager_path','default_path','rccl_1rank'): print(k, {a: b for a, b in d.get(k, {}).items() if a in ('value','ms_per_step','launch','error')})"
