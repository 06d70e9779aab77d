// monodetr_amd/csrc/conv_taps.h -- internal launcher declarations (see conv_taps.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// y[b, r, c, n] = act(shift[n] + sum_{a < TR, e < TS, k < C} x[b, SI r + a - PT, SI c + e - PL, k] * w[n, ta0 + a ta_step, te0 + e te_step, k])
// for r < OH, c < OW; x [B, H, W, C] bf16 contiguous; the output pixel at y + y_off + b y_sb + r y_sr + c y_sc (elements);
// the weight element (n, tap row, tap column, k) at w + n w_sn + row w_sa + column w_se + k.
struct ConvTapsDims {
    int B, H, W, C;
    int OH, OW, N;
    int SI, TR, TS, PT, PL;
    int ta0, ta_step, te0, te_step;
    int64_t y_off, y_sb, y_sr, y_sc;
    int64_t w_sn, w_sa, w_se;
};

bool conv_taps_supported(const ConvTapsDims &d, const void *x, const void *w, const void *y);
// the input gradient of a 3x3 (pad 1) or 1x1 (pad 0) stride-2 convolution, all four pixel-parity classes in one launch:
// dy [B, OH, OW, N] (N % 64 == 0), wt [C, K, K, N] (channel axes swapped, taps not mirrored), dx [B, H, W, C] (C % 32 == 0)
bool conv_dgrad_s2_supported(int B, int OH, int OW, int N, int H, int W, int C, int K, const void *dy, const void *wt, const void *dx);
hipError_t conv_dgrad_s2_launch(const void *dy, const void *wt, void *dx, int B, int OH, int OW, int N, int H, int W, int C, int K, hipStream_t st);
// split over the contraction channels: fp32 partials [ksplit][B][OH][OW][N] (split 0 carries the shift), summed by the caller
bool conv_taps_split_supported(const ConvTapsDims &d, int ksplit);
hipError_t conv_taps_split_launch(const void *x, const void *w, const float *shift, float *part, const ConvTapsDims &d, int ksplit, hipStream_t st);
hipError_t conv_taps_launch(const void *x, const void *w, const float *shift, void *y, const ConvTapsDims &d, bool relu, hipStream_t st);

}  // namespace mdetr
