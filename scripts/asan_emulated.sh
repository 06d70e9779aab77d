#!/bin/bash
# CPU-only checks of the kernel sources beyond the default test run (tests/native_emul.py builds them on the HIP-on-CPU shim):
#   1. AddressSanitizer build: an out-of-bounds global or LDS access in any kernel aborts with the kernel's file:line;
#   2. lane-order permutations (HIPSHIM_ORDER=reverse / shuffle): between two synchronisation points the shim runs a
#      block's lanes one after another -- in another order a consumer that is not separated from its producer by a
#      barrier runs first and the result changes.
#   ./scripts/asan_emulated.sh            (~8 min on 8 vCPUs)
cd "$(dirname "$0")/.."
T="tests/test_msda_emulated_cpu.py tests/test_attn_emulated_cpu.py tests/test_add_ln_emulated_cpu.py tests/test_tgemm_emulated_cpu.py tests/test_twgrad_emulated_cpu.py tests/test_wfold_emulated_cpu.py
   tests/test_kitti_eval_cpu.py tests/test_msda_prologue_cpu.py tests/test_bias_act_emulated_cpu.py tests/test_conv3x3_emulated_cpu.py tests/test_lsa_emulated_cpu.py tests/test_fused_losses_cpu.py tests/test_optimizer.py tests/test_kitti_pipeline_cpu.py"
set -e
for order in reverse shuffle; do
    echo "== lane order: $order"
    HIPSHIM_ORDER=$order python -m pytest -q -x -k "not install" $T "$@" | tail -1
done
echo "== AddressSanitizer"
MDETR_EMUL_ASAN=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
    python -m pytest -q -x -k "not install" $T tests/test_step_emulated_cpu.py "$@" | tail -1
