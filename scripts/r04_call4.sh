#!/bin/bash
# Round 4, call 4: cycles per phase of the fused MSDA backward (developer build -DMDETR_PHASES)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04d; mkdir -p $O
cd $R
export TMPDIR=/tmp
MDETR_LIB_PATH=$R/monodetr_amd/variants/lib_phases.so timeout 120 python -m monodetr_amd.tools.opbench --dtype bf16 --dist init --iters 20 --phases 2>&1 | tee $O/phases.log | cut -c1-600
