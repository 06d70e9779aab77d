#!/bin/bash
# tests named in $TESTS (default: the fused-kernel and model GPU tests), then one bench line without side legs
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${TAG:-check}; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 1200 python -m pytest ${TESTS:-tests/test_fused_gpu.py tests/test_model_gpu.py} -x -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log; grep -n "^E  \|^FAILED" $O/pytest.log | cut -c1-300 | head -8
TAG=${TAG:-check} bash scripts/r04_bench.sh
