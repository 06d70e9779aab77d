#!/bin/bash
# Round 2, GPU call 27: tools/train_val.py end to end with the committed kernel families switched on by default.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02z; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 170 python -X faulthandler -m pytest tests/test_fused_gpu.py -x -q -s -p no:cacheprovider --timeout 160 -k "train_val_entry_point" > $O/pytest_e2e.log 2>&1; echo "rc=$?"; grep -n "passed\|failed\|Error\|Kernel families" $O/pytest_e2e.log | tail -6 | cut -c1-300
