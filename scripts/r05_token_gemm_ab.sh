#!/bin/bash
# Round 5, first call: the committed step against the same list + MDETR_TOKEN_GEMM with the weight-in-registers form
# (MDETR_TOKEN_GEMM_DIRECT=2) -- same box, two repetitions each, then the GPU tests that cover the routed layers
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05tg; mkdir -p $O
cd $R
LIST=$(python -c "import bench; print(' '.join(k + '=1' for k in sorted(bench.COMMITTED_SWITCHES['bf16'])))" 2>/dev/null)
run() { env "$@" timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-variants 2>$O/$TAGN.err | tail -1 > $O/$TAGN.json
  python -c "import json; d=json.loads(open('$O/$TAGN.json').read()); print('$TAGN', d['value'], d['ms_per_step'], d['final_loss'], d['config']['switch_source'])"; }
for rep in 1 2; do
  TAGN=committed_$rep run $LIST
  TAGN=token_gemm_regs_$rep run $LIST MDETR_TOKEN_GEMM=1 MDETR_TOKEN_GEMM_DIRECT=2
  TAGN=token_gemm_lds_$rep run $LIST MDETR_TOKEN_GEMM=1 MDETR_TOKEN_GEMM_DIRECT=0
done
timeout 300 python -m pytest tests/test_fused_gpu.py -m gpu -q -k "token_gemm or token_linear" -p no:cacheprovider 2>&1 | tail -3
