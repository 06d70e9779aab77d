#!/bin/bash
# Round 3, call 4: experiments on the non-finite gradients of the replayed iteration (serialised queues, fresh small_wgrad
# outputs, a family left out, quiet replays with host sleeps, the graph's node types), the convolutions still in the library.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 "$@" 2>&1 | grep -v Warning | grep "^i=\|twin\|Error\|error\|grad \|^   " | cut -c1-700 | tail -30; }
{
run python tests/diag/graph_nan.py
run env AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 python tests/diag/graph_nan.py --no-ref
run python tests/diag/graph_nan.py --no-ref --fresh-wgrad
run python tests/diag/graph_nan.py --no-ref --drop MDETR_SMALL_WGRAD
run python tests/diag/graph_nan.py --no-ref --drop MDETR_FUSED_ADAMW
run python tests/diag/graph_nan.py --no-ref --quiet --steps 30 --lr 2e-4
run python tests/diag/graph_nan.py --no-ref --quiet --sleep 0.05 --steps 30 --lr 2e-4
run python tests/diag/graph_nan.py --no-ref --steps 5 --dump $O/graph
} > $O/graph_nan.log 2>&1
cat $O/graph_nan.log
for f in $O/graph.*.dot; do echo $f; grep -o 'label="[A-Za-z_ ]*' $f | sort | uniq -c | sort -rn | head -8; grep -c -i "memcpy" $f; grep -c -i "memset" $f; done 2>/dev/null | head -40
rm -f $O/graph.*.dot
timeout 600 python -m pytest tests/test_fused_gpu.py -q -p no:cacheprovider --timeout 600 -k "convolution_kernels" > $O/pytest_conv.log 2>&1; echo "pytest conv rc=$?"
cat gpurun_out/library_convolutions.txt 2>/dev/null | cut -c1-200 | head -40
