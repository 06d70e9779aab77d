// monodetr_amd/csrc/colsum.h -- internal launcher declarations (see colsum.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// dtype: 0 = f32, 2 = bf16 (MDETR_F32 / MDETR_BF16 of include/monodetr_amd.h)
bool colsum_supported(int dtype, int cols, int64_t ld, const void *x);
int64_t colsum_workspace_bytes(int64_t rows, int cols);
hipError_t colsum_launch(int dtype, const void *x, float *out, void *workspace, int64_t rows, int cols, int64_t ld,
                         hipStream_t st);

}  // namespace mdetr
