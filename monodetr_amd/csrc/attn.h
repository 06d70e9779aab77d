// monodetr_amd/csrc/attn.h -- internal launcher declarations of the fused attention kernels (attn.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

struct AttnProblem {
    int dtype;                      // 0 = fp32 I/O, 2 = bf16 I/O (MDETR_F32 / MDETR_BF16); math is bf16 MFMA + fp32 accumulate
    const void *q, *k, *v;          // [B, L, H*32], row strides in elements, innermost contiguous
    const uint8_t *key_padding_mask;   // [B, Lk] nonzero = ignore, or null
    int B, H, Lq, Lk;
    int64_t q_batch_stride, k_batch_stride, v_batch_stride;
    int q_row_stride, k_row_stride, v_row_stride;
    float scale, dropout_p;
    uint64_t seed;
    const uint64_t *seed_dev;        // optional device word added to `seed` (graph-replay-safe dropout), or null
};

hipError_t attn_forward_launch(const AttnProblem &p, void *out, float *lse2, hipStream_t st);
hipError_t attn_backward_launch(const AttnProblem &p, const void *out, const void *d_out, const float *lse2,
                                float *dsum, void *dq, void *dk, void *dv, hipStream_t st);

}  // namespace mdetr
