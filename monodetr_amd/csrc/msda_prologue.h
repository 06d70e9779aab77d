// monodetr_amd/csrc/msda_prologue.h -- internal launcher declarations (see msda_prologue.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

struct PrologueDims {
    int B, Lq, M, L, P, R;               // R = 2 or 6 reference components
    int64_t rsb, rsq, rsl;               // element strides of the reference points [B, Lq, L, R] (last dim contiguous)
    int64_t po = 0, pl = 0;              // elements from one (b, q) row of the offsets / logits to the next; 0 = dense (M L P 2, M L P).
                                         // A PACKED projection output [B, Lq, M L P 3] (offsets | logits of one GEMM) has po = pl = M L P 3
                                         // and its logits start M L P 2 elements into the row (L = P = 4 form only)
};

// io_dtype: 0 = f32, 2 = bf16 (offsets, logits and the returned g_offsets / g_logits); ref_dtype: the reference points'
hipError_t msda_prologue_forward_launch(int io_dtype, int ref_dtype, const PrologueDims &d, const void *offsets, const void *logits,
                                        const void *ref, const int64_t *shapes, float *loc, float *attn, hipStream_t st);
// g_ref: fp32 [B, Lq, L, R] dense, accumulated with atomics -- the caller zero-fills it; NULL = not needed
hipError_t msda_prologue_backward_launch(int io_dtype, int ref_dtype, const PrologueDims &d, const void *offsets, const void *ref,
                                         const int64_t *shapes, const float *attn, const float *g_loc, const float *g_attn,
                                         void *g_offsets, void *g_logits, float *g_ref, hipStream_t st);

}  // namespace mdetr
