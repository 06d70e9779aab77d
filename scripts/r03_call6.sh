#!/bin/bash
# Round 3, call 6: the replay corruption with the runtime's own defaults, on the caller's stream (control) and on the graphs' own
# stream; the product's defaults (packet path off + own stream); what the packet path is worth in the benchmark; the trainer /
# convolution tests on the fixed tree.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 "$@" 2>&1 | grep -v Warning | grep "^i=\|Error\|error\|grad \|^   " | cut -c1-260 | tail -9; }
{
run env MDETR_DIAG_RUNTIME_DEFAULTS=1 MDETR_REPLAY_STREAM=current python tests/diag/graph_nan.py --no-ref
run env MDETR_DIAG_RUNTIME_DEFAULTS=1 MDETR_REPLAY_STREAM=own python tests/diag/graph_nan.py --no-ref
run python tests/diag/graph_nan.py
} > $O/graph_nan.log 2>&1
cat $O/graph_nan.log
for pc in 1 0; do
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=$pc timeout 300 python bench.py --no-variants --no-cpu-baseline --steps 40 2>$O/bench_pc$pc.err | tail -1 > $O/bench_pc$pc.json
  python -c "import json; d=json.load(open('$O/bench_pc$pc.json')); print('packet capture $pc:', d['value'], d['ms_per_step'], d['final_loss'], d['config']['launch'][:24])" || tail -3 $O/bench_pc$pc.err
done
timeout 1500 python -m pytest tests/test_trainer_gpu.py tests/test_graph_gpu.py "tests/test_fused_gpu.py::test_train_val_entry_point_end_to_end" -q -p no:cacheprovider --timeout 900 -s > $O/pytest_trainer.log 2>&1; echo "pytest trainer rc=$?"
grep -n "passed\|failed\|PG-CHILD\|spread\|worst\|^E  " $O/pytest_trainer.log | cut -c1-400 | tail -14
timeout 900 python -m pytest tests/test_fused_gpu.py -q -p no:cacheprovider --timeout 600 -k "conv_strided or conv_wgrad or conv_stem or convolution_kernels" > $O/pytest_conv.log 2>&1; echo "pytest conv rc=$?"
grep -n "passed\|failed\|^E  " $O/pytest_conv.log | cut -c1-1200 | tail -6
cat gpurun_out/library_convolutions.txt 2>/dev/null | cut -c1-160 | head -20
