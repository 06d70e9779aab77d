#!/bin/bash
# Round 3, call 5: what makes the replayed iteration's gradients non-finite when eager work runs between replays -- an observation
# race (settle time), the runtime's graph packet capture, SDMA copies, hardware queues.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 "$@" 2>&1 | grep -v Warning | grep "^i=\|Error\|error\|grad \|^   " | cut -c1-330 | tail -14; }
{
run python tests/diag/graph_nan.py --no-ref --settle 0.5
run env DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tests/diag/graph_nan.py --no-ref
run env AMD_SERIALIZE_KERNEL=3 python tests/diag/graph_nan.py --no-ref
run env AMD_SERIALIZE_COPY=3 python tests/diag/graph_nan.py --no-ref
run env HSA_ENABLE_SDMA=0 python tests/diag/graph_nan.py --no-ref
run env GPU_MAX_HW_QUEUES=1 python tests/diag/graph_nan.py --no-ref
} > $O/graph_nan.log 2>&1
cat $O/graph_nan.log
