// monodetr_amd/csrc/adamw.h -- internal launcher declaration (see adamw.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// param_dtype: 0 = f32 (master == param), 2 = bf16 (master = fp32 copy, param = rounded model copy)
hipError_t adamw_launch(int param_dtype, void *param, float *master, const void *grad, float *exp_avg,
                        float *exp_avg_sq, int64_t n, int64_t n_no_decay, float beta1, float beta2, float eps,
                        float weight_decay, float step_host, const float *step_dev, hipStream_t st,
                        const double *count_dev = nullptr, const double *lr_dev = nullptr, float lr_host = 0.f);

}  // namespace mdetr
