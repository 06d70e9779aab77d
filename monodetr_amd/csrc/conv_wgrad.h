// monodetr_amd/csrc/conv_wgrad.h -- internal launcher declarations (see conv_wgrad.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// x [B, H, W, C] bf16, dy [B, OH, OW, N] bf16 (both contiguous, 16-byte aligned), K x K taps (K = 3: pad 1, stride SI in {1, 2};
// K = 1: pad 0, stride 2), C % 64 == 0, N % 32 == 0.  part: fp32 [chunks][N][K][K][C], chunks = conv_wgrad_chunks(d).
struct ConvWgradDims {
    int B, H, W, C;
    int OH, OW, N;
    int K, SI;
    int DB = 0;       // 1 (K = 1, SI = 1 only): every chunk also carries the column sums of dy -- the bias gradient of a token-wise linear
                      // layer -- as N more floats behind its [N][C] block: part is [chunks][N * C + N]
};

bool conv_wgrad_supported(const ConvWgradDims &d, const void *x, const void *dy);
int conv_wgrad_chunks(const ConvWgradDims &d);
hipError_t conv_wgrad_launch(const void *x, const void *dy, float *part, const ConvWgradDims &d, hipStream_t st);

}  // namespace mdetr
