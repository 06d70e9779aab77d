// monodetr_amd/csrc/token_gemm.h -- internal launcher declarations (see token_gemm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// bf16 only; K in {64, 128, 256, 512}; N % 8 == 0; x rows 16-byte aligned (ldx % 8 == 0), y rows 8-byte aligned
bool token_gemm_supported(int64_t T, int N, int K, int64_t ldx, int64_t ldy, const void *x, const void *w, const void *y);
hipError_t token_gemm_launch(const void *x, const void *w, const void *bias, void *y, int64_t T, int N, int K,
                             int64_t ldx, int64_t ldy, bool relu, hipStream_t st);

}  // namespace mdetr
