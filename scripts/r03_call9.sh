#!/bin/bash
# Round 3, call 9: operator map of the eager iteration (which operator issues the framework launches) and the bf16-vs-fp32 gradient
# cosines with fp32 values in the decoder's cross-attention.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03j; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 400 python -m monodetr_amd.tools.opmap --top 500 --out $O/opmap.txt > /dev/null 2>$O/opmap.err; head -40 $O/opmap.txt | cut -c1-200; tail -2 $O/opmap.err
timeout 400 python -m pytest tests/test_model_gpu.py -x -q -s -m gpu -p no:cacheprovider -k bf16_body > $O/bf16_body.log 2>&1; tail -2 $O/bf16_body.log; grep -n "cosine\|below" $O/bf16_body.log | cut -c1-1500
