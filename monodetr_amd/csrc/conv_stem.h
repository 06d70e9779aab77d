// monodetr_amd/csrc/conv_stem.h -- internal launcher declarations (see conv_stem.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// x [B, H, W, 3] bf16, wp: packed weight [64][176] bf16 (element t * 24 + e * 3 + ch of row n = w[n, ch, t, e]; the rest zero),
// shift fp32 [64] or null, y [B, (H - 1) / 2 + 1, (W - 1) / 2 + 1, 64] bf16 = relu(conv7x7 stride 2 pad 3 + shift)
bool conv_stem_supported(int B, int H, int W, const void *x, const void *wp, const void *y);
hipError_t conv_stem_launch(const void *x, const void *wp, const float *shift, void *y, int B, int H, int W, hipStream_t st);

}  // namespace mdetr
