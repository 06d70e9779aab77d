// monodetr_amd/csrc/msda_prologue_math.h -- what MSDeformAttn.forward does between its two projections and the
// sampling operator (lib/models/monodetr/ops/modules/ms_deform_attn.py:139-160): softmax of the attention
// logits over the L*P samples of a head, and sampling locations from reference points + offsets; values and
// gradients.  Shared by the HIP kernels (msda_prologue.hip) and the host build of the CPU tests.
//
// Work unit = one (image, query, head): LP = L * P <= 64 samples.
#pragma once

#if defined(__HIPCC__)
#define MDETR_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define MDETR_HD inline
#endif

namespace mdetr {

constexpr int kPrologueMaxLP = 64;

MDETR_HD float pro_exp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __expf(x);
#else
    return std::exp(x);
#endif
}

// attn = softmax(logit[0 .. LP))
MDETR_HD void pro_softmax(const float *logit, int LP, float *attn)
{
    float mx = logit[0];
    for (int i = 1; i < LP; ++i) mx = logit[i] > mx ? logit[i] : mx;
    float se = 0.f;
    for (int i = 0; i < LP; ++i) { attn[i] = pro_exp(logit[i] - mx); se += attn[i]; }
    const float inv = 1.f / se;
    for (int i = 0; i < LP; ++i) attn[i] *= inv;
}

// g_logit = attn * (g_attn - sum_i attn_i g_attn_i)
MDETR_HD void pro_softmax_backward(const float *attn, const float *g_attn, int LP, float *g_logit)
{
    float dot = 0.f;
    for (int i = 0; i < LP; ++i) dot += attn[i] * g_attn[i];
    for (int i = 0; i < LP; ++i) g_logit[i] = attn[i] * (g_attn[i] - dot);
}

// sampling location of sample (l, p), component c (0 = x, 1 = y):
//   R == 2 (:147-151):  ref[l][c] + off / (W_l, H_l)[c]
//   R == 6 (:152-155):  ref[l][c] + off / P * (ref[l][2 + 2c] + ref[l][3 + 2c]) * 0.5
MDETR_HD float pro_location(float off, const float *ref_l, int R, int c, float wh_c, int P)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    if (R == 2) {
        const float d = off / wh_c;
        return ref_l[c] + d;
    }
    const float extent = ref_l[2 + 2 * c] + ref_l[3 + 2 * c];
    const float d = off / static_cast<float>(P) * extent * 0.5f;
    return ref_l[c] + d;
}

// d loc / d off, and this sample's contribution to d loss / d ref_l (R values, accumulated)
MDETR_HD float pro_location_backward(float g_loc, float off, const float *ref_l, int R, int c, float wh_c, int P, float *g_ref_l)
{
    if (R == 2) {
        if (g_ref_l) g_ref_l[c] += g_loc;
        return g_loc / wh_c;
    }
    const float extent = ref_l[2 + 2 * c] + ref_l[3 + 2 * c];
    if (g_ref_l) {
        g_ref_l[c] += g_loc;
        const float ge = g_loc * (off / static_cast<float>(P)) * 0.5f;
        g_ref_l[2 + 2 * c] += ge;
        g_ref_l[3 + 2 * c] += ge;
    }
    return g_loc / static_cast<float>(P) * extent * 0.5f;
}

}  // namespace mdetr
