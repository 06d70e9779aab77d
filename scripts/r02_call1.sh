#!/bin/bash
# Round 2, GPU call 1: first GPU run of every kernel family written after round 1's budget ran out (the 57 tests of
# tests/test_pending_gpu.py), op-level timings of the same kernels, and the step with / without them.
#   gpurun --timeout 1500 -- './scripts/r02_call1.sh'
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02a; mkdir -p $O
cd $R
export TMPDIR=/tmp
MDETR_TEST_PENDING=1 timeout 600 python -m pytest tests/test_pending_gpu.py -q -rA -p no:cacheprovider --timeout 240 > $O/pytest_pending.log 2>&1
tail -80 $O/pytest_pending.log | grep -E "^(PASSED|FAILED|ERROR|SKIPPED)|passed|failed" | head -90
timeout 240 python -m monodetr_amd.tools.fusedbench > $O/fusedbench.json 2> $O/fusedbench.err; tail -c 600 $O/fusedbench.err
timeout 200 python -m monodetr_amd.tools.prepbench > $O/prepbench_fp32.json 2> $O/prepbench.err; tail -c 400 $O/prepbench.err
timeout 200 python -m monodetr_amd.tools.prepbench --dtype bf16 > $O/prepbench_bf16.json 2>> $O/prepbench.err
timeout 300 python -m monodetr_amd.tools.evalbench > $O/evalbench.json 2> $O/evalbench.err; tail -c 400 $O/evalbench.err
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %7.2f img/s %7.3f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
run() { env MDETR_BENCH_AUTOTUNE=0 $1 timeout 240 python bench.py --no-cpu-baseline 2>$O/bench_err_$2.log | tee $O/bench_$2.json | val "${1:-default}"; }
run "" default
run "MDETR_FUSED_LOSSES=1 MDETR_FUSED_ADAMW=1 MDETR_FUSED_LN=1 MDETR_MSDA_PROLOGUE=1 MDETR_MSDA_BF16=1 MDETR_FUSED_EPILOGUE=1 MDETR_GEMM_RELU=1" seven
run "MDETR_FUSED_LOSSES=1 MDETR_FUSED_ADAMW=1" crit_opt
run "MDETR_CONV3X3=1" conv3x3
run "MDETR_TOKEN_GEMM=1" token_gemm
env MDETR_BENCH_AUTOTUNE=0 timeout 240 python bench.py --precision fp32 --no-cpu-baseline 2>$O/bench_err_fp32.log | tee $O/bench_fp32.json | val "fp32 default"
ls -la $O
