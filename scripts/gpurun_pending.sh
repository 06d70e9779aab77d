#!/bin/bash
# First GPU run of the next round: validate the kernels that were written after the round-1 GPU budget was spent
# (DESIGN.md 7.0), then measure each switch on its own and all together.
#   gpurun --timeout 1500 -- './scripts/gpurun_pending.sh'
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pending; mkdir -p $O
cd $R
MDETR_TEST_PENDING=1 python -m pytest tests/test_pending_gpu.py -q -x 2>&1 | tail -15 | tee $O/pytest_pending.log
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %7.2f img/s %7.3f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
run() { env MDETR_BENCH_AUTOTUNE=0 $1 python bench.py --no-cpu-baseline 2>/dev/null | val "${1:-default}"; }
run ""
run "MDETR_FUSED_LOSSES=1"
run "MDETR_FUSED_ADAMW=1"
run "MDETR_MSDA_PROLOGUE=1"
run "MDETR_TOKEN_GEMM=1"
run "MDETR_MSDA_BF16=1"
run "MDETR_FUSED_LN=1"
run "MDETR_FUSED_EPILOGUE=1"
run "MDETR_GEMM_RELU=1"
run "MDETR_FUSED_EPILOGUE=1 MDETR_GEMM_RELU=1"
run "MDETR_CONV3X3=1"
run "MDETR_BENCH_MIOPEN_FIND=1"
run "MDETR_BENCH_TUNABLEOP=1"
run "MDETR_FUSED_LOSSES=1 MDETR_FUSED_ADAMW=1 MDETR_MSDA_PROLOGUE=1 MDETR_TOKEN_GEMM=1 MDETR_MSDA_BF16=1 MDETR_FUSED_LN=1 MDETR_FUSED_EPILOGUE=1 MDETR_GEMM_RELU=1 MDETR_CONV3X3=1"
env MDETR_BENCH_AUTOTUNE=0 MDETR_FUSED_LOSSES=1 MDETR_FUSED_ADAMW=1 MDETR_MSDA_PROLOGUE=1 python bench.py --precision fp32 --no-cpu-baseline 2>/dev/null | val "fp32, fused losses + adamw + prologue"
python -m monodetr_amd.tools.prepbench 2>&1 | tail -1 | tee $O/prepbench_fp32.json
python -m monodetr_amd.tools.prepbench --dtype bf16 2>&1 | tail -1 | tee $O/prepbench_bf16.json
python tests/prep_cpu_baseline.py | tee $O/prep_cpu_baseline.json
python -m monodetr_amd.tools.evalbench 2>&1 | tail -1 | tee $O/evalbench.json
rm -f ${TMPDIR:-/tmp}/mdetr_bench_autotune.json; python bench.py --no-cpu-baseline 2>$O/bench_autotune.err | tee $O/bench_autotune.json | val "start-up probe (default invocation)"
python -m monodetr_amd.tools.fusedbench 2>&1 | tail -1 | tee $O/fusedbench.json
