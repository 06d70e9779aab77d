// monodetr_amd/csrc/conv3x3.h -- internal launcher declarations (see conv3x3.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// bf16 only; x [B, H, W, C] with C % 64 == 0, w [N, 3, 3, C] with N % 32 == 0, y [B, H, W, N]; x / w 16-byte, y 8-byte aligned
bool conv3x3_supported(int B, int H, int W, int C, int N, const void *x, const void *w, const void *y);
// mirror: tap (t, s) reads w[n][2 - t][2 - s][:] (the input gradient on the weight with swapped channel axes)
// mask (bf16 [B, H, W, N], 8-byte aligned, or null): the result is zeroed where mask <= 0
hipError_t conv3x3_launch(const void *x, const void *w, const float *shift, void *y, int B, int H, int W, int C, int N, bool relu,
                          hipStream_t st, bool mirror = false, const void *mask = nullptr);

// the launcher's choice for a shape: 100 x (columns of a wave's pixel block) + 10 x (blocks side by side) + output-channel blocks of 32
int conv3x3_plan(int B, int H, int W, int N);

}  // namespace mdetr
