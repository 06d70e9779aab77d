// tests/native/host_kernels.cpp -- HOST build (g++) of the per-element arithmetic the HIP kernels use,
// so that CPU tests can validate it against recorded reference results without a GPU.  Test
// infrastructure only: nothing in the product links or loads this.
#include <stdint.h>
#include <string.h>

#include "../../monodetr_amd/csrc/adamw_math.h"

namespace {

inline float bf16_to_f32(uint16_t h)
{
    uint32_t u = static_cast<uint32_t>(h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline uint16_t f32_to_bf16(float f)                    // round to nearest even, as __float2bfloat16
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return static_cast<uint16_t>((u >> 16) | 0x40);   // NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return static_cast<uint16_t>(u >> 16);
}

}  // namespace

extern "C" {

// same argument list as mdetr_adamw_step (include/monodetr_amd.h); device / stream are ignored
int mdetr_adamw_step(int param_dtype, void *param, float *master, const void *grad, float *exp_avg, float *exp_avg_sq,
                     int64_t n, int64_t n_no_decay, float beta1, float beta2, float eps, float weight_decay,
                     float step_size, const float *step_size_dev, int device, void *stream)
{
    (void)device; (void)stream;
    const mdetr::AdamWCoef c{beta1, beta2, eps, weight_decay};
    const float step = step_size_dev ? *step_size_dev : step_size;
    for (int64_t i = 0; i < n; ++i) {
        const float g = param_dtype == 2 ? bf16_to_f32(static_cast<const uint16_t *>(grad)[i])
                                         : static_cast<const float *>(grad)[i];
        float m = exp_avg[i], v = exp_avg_sq[i];
        const float q = mdetr::adamw_element(master[i], g, m, v, c, i < n_no_decay ? 0.f : weight_decay, step);
        exp_avg[i] = m; exp_avg_sq[i] = v; master[i] = q;
        if (param_dtype == 2) static_cast<uint16_t *>(param)[i] = f32_to_bf16(q);
    }
    return 0;
}

}  // extern "C"
