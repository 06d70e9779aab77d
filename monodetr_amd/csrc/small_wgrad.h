// monodetr_amd/csrc/small_wgrad.h -- internal launcher declarations (see small_wgrad.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// io_dtype / out_dtype: 0 = f32, 2 = bf16
bool small_wgrad_supported(int io_dtype, int64_t rows, int n, int k, int64_t ldy, int64_t ldx);
int small_wgrad_chunks(int64_t rows, int n, int k);
int64_t small_wgrad_workspace_bytes(int64_t rows, int n, int k);
// out: [n * k + n] in out_dtype: dW (row-major [n, k]) followed by db
hipError_t small_wgrad_launch(int io_dtype, const void *dy, const void *x, void *out, void *workspace, int64_t rows, int n, int k,
                              int64_t ldy, int64_t ldx, int out_dtype, hipStream_t st);

}  // namespace mdetr
