// monodetr_amd/csrc/twgrad.h -- internal launcher declarations (see twgrad.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// bf16 x [T, C] (row stride ldx), dy [T, N] (row stride ldy); C % 8 == 0, N % 8 == 0; 16-byte aligned
bool twgrad_supported(int64_t T, int C, int N, int64_t ldx, int64_t ldy, const void *x, const void *dy);
int twgrad_chunks(int64_t T, int C, int N);
// part: fp32 [chunks][N * C (+ N with_db)]: per-chunk partial dW (row n, column c) followed by the chunk's partial db
hipError_t twgrad_launch(const void *x, const void *dy, float *part, int64_t T, int C, int N, int64_t ldx, int64_t ldy, bool with_db,
                         hipStream_t st);

}  // namespace mdetr
