#!/bin/bash
# Round 2, GPU call 2: the new parity tests (bf16 body vs fp32 golden, bf16-native MSDA at the full shape vs the oracle,
# FusedAdamW vs recorded reference steps, entry point), the restructured bench.py (committed switches, side lines, CPU
# baseline protocol), configs 2 and 5, and the committed list + the 3x3 convolution kernel.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02b; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_msda_gpu.py::test_bf16_native_full_encoder_shape_vs_oracle tests/test_fused_gpu.py::test_fused_adamw_vs_recorded_reference_steps tests/test_fused_gpu.py::test_train_val_entry_point_end_to_end -q -rA -s -p no:cacheprovider --timeout 400 > $O/pytest_new.log 2>&1
grep -E "^(PASSED|FAILED|ERROR)|passed|failed|lowest gradient|worst relative" $O/pytest_new.log | head -40
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %7.2f img/s %7.3f ms' % (sys.argv[1], d['value'], d['ms_per_step'])); [print('   ', k, d[k].get('value'), d[k].get('ms_per_step', '')) for k in ('fp32_path','default_path','rccl_1rank','cpu_baseline') if k in d]" "$1"; }
( time timeout 900 python bench.py 2>$O/bench_err_committed.log | tee $O/bench_committed.json | val "committed (full line)" ) 2>&1 | grep -v "^$" | grep -v user | grep -v sys
env MDETR_FUSED_LOSSES=1 MDETR_FUSED_ADAMW=1 MDETR_FUSED_LN=1 MDETR_MSDA_PROLOGUE=1 MDETR_MSDA_BF16=1 MDETR_FUSED_EPILOGUE=1 MDETR_GEMM_RELU=1 MDETR_CONV3X3=1 timeout 300 python bench.py --no-cpu-baseline --no-variants 2>$O/bench_err_conv.log | tee $O/bench_committed_plus_conv3x3.json | val "committed + conv3x3"
timeout 300 python bench.py --config 2 --no-cpu-baseline --no-variants 2>$O/bench_err_c2.log | tee $O/bench_config2.json | val "config 2 (backbone+encoder fp32)"
timeout 400 python bench.py --config 5 --no-cpu-baseline --no-variants 2>$O/bench_err_c5.log | tee $O/bench_config5.json | val "config 5 (512x1760, 100 q, bf16)"
tail -5 $O/bench_err_c2.log $O/bench_err_c5.log $O/bench_err_committed.log | grep -v amdgpu.ids
