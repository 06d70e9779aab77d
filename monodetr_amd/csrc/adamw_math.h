// monodetr_amd/csrc/adamw_math.h -- one element of the reference's AdamW update
// (lib/helpers/optimizer_helper.py:104-129), shared by the HIP kernel (adamw.hip) and by the host build
// the CPU tests compile with g++ (tests/native/host_kernels.cpp), so the arithmetic is validated
// against the recorded reference trajectory without a GPU.
//
//   m <- b1 m + (1 - b1) g ;  v <- b2 v + (1 - b2) g^2
//   p <- p - step * (wd * p + m / (sqrt(v) + eps)),   step = lr * sqrt(1 - b2^t) / (1 - b1^t)
//
// i.e. NOT torch.optim.AdamW: the decoupled decay is scaled by the bias-corrected step and eps is
// added outside the bias correction.
#pragma once

#if defined(__HIPCC__)
#define MDETR_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define MDETR_HD inline
#endif

namespace mdetr {

struct AdamWCoef { float beta1, beta2, eps, weight_decay; };

// updates m, v in place and returns the new parameter value
MDETR_HD float adamw_element(float p, float g, float &m, float &v, const AdamWCoef &c, float wd, float step)
{
    m = c.beta1 * m + (1.0f - c.beta1) * g;
    v = c.beta2 * v + (1.0f - c.beta2) * g * g;
#if defined(__HIP_DEVICE_COMPILE__)
    const float denom = __fsqrt_rn(v) + c.eps;
#else
    const float denom = std::sqrt(v) + c.eps;
#endif
    return p - step * (wd * p + m / denom);
}

}  // namespace mdetr
