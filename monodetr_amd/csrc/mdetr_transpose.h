// monodetr_amd/csrc/mdetr_transpose.h -- an 8 x 8 bf16 block transposed in registers (8 rows of 8 values in, 8 columns of 8
// values out): what an MFMA operand whose contraction index is the SLOW axis of its tensor needs on the way into LDS
// (conv_wgrad.hip: the pixel axis; tgemm.hip: the input-gradient form, whose weight lies [n][k_in]).
#pragma once
#include <mdetr_wave.h>

namespace mdetr {

__device__ __forceinline__ void transpose8x8(const bf16x8 (&in)[8], bf16x8 (&out)[8])
{
    // in[i] = 8 channels of row i; out[q] = 8 rows of channel q.  Word m of out[q] = (in[2m][q], in[2m + 1][q]).
    unsigned d[8][4], o[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_memcpy(d[i], &in[i], 16);
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const unsigned lo = d[2 * m][q >> 1], hi = d[2 * m + 1][q >> 1];
            o[q][m] = (q & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        }
#pragma unroll
    for (int q = 0; q < 8; ++q) __builtin_memcpy(&out[q], o[q], 16);
}

}  // namespace mdetr
