#!/bin/bash
# Memory-safety pass over the kernel sources: the HIP-on-CPU emulation build (tests/native_emul.py) compiled with
# AddressSanitizer, so an out-of-bounds global or LDS access in any kernel aborts with the kernel's file:line.
#   ./scripts/asan_emulated.sh            (CPU only, ~5 min)
cd "$(dirname "$0")/.."
export MDETR_EMUL_ASAN=1 LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0
exec python -m pytest -q -x -k "not install" tests/test_msda_emulated_cpu.py tests/test_attn_emulated_cpu.py tests/test_add_ln_emulated_cpu.py \
    tests/test_token_gemm_emulated_cpu.py tests/test_kitti_eval_cpu.py tests/test_msda_prologue_cpu.py tests/test_fused_losses_cpu.py \
    tests/test_optimizer.py tests/test_kitti_pipeline_cpu.py tests/test_step_emulated_cpu.py "$@"
