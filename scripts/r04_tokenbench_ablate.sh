#!/bin/bash
# Where the weight-in-registers token GEMM spends its time: the encoder shape with stores / memory reads / products removed
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04tga; mkdir -p $O
cd $R; timeout 80 python -m monodetr_amd.tools.tokenbench --iters 40 --ablate --only "${ONLY:-encoder_256to256}" --out $O/tokenbench_ablate.json 2>&1 | tail -3 | cut -c1-900
