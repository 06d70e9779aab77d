#!/bin/bash
# Round 2, GPU call 28: the end-to-end entry-point test (committed kernel families on, switches reset afterwards) followed in
# the same process by the model tests that expect the default switches.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02z; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_fused_gpu.py tests/test_model_gpu.py -x -q -p no:cacheprovider --timeout 180 -k "train_val_entry_point or test_model_gpu" > $O/pytest_e2e_then_model.log 2>&1; echo "rc=$?"; grep -n "passed\|failed\|Error" $O/pytest_e2e_then_model.log | tail -4 | cut -c1-300
