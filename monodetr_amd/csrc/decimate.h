// monodetr_amd/csrc/decimate.h -- internal launcher declarations (see decimate.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// pixel_bytes = C * element size, a multiple of 16; x [B, H, W, C] channels-last, y [B, (H + 1) / 2, (W + 1) / 2, C]
bool decimate2_supported(int64_t pixel_bytes, const void *x, const void *y);
hipError_t decimate2_forward_launch(const void *x, void *y, int B, int H, int W, int64_t pixel_bytes, hipStream_t st);
// dx [B, H, W, C]: dy at the even pixels, zero elsewhere (every byte of dx is written)
hipError_t decimate2_backward_launch(const void *dy, void *dx, int B, int H, int W, int64_t pixel_bytes, hipStream_t st);

// 3x3 / stride 2 / pad 1 max pooling, bf16 channels-last, C a multiple of 8: x [B, H, W, C] -> y [B, (H - 1) / 2 + 1, (W - 1) / 2 + 1, C]
hipError_t maxpool3x3s2_bf16_launch(const void *x, void *y, int B, int H, int W, int C, hipStream_t st);

}  // namespace mdetr
