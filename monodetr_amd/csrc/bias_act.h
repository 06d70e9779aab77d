// monodetr_amd/csrc/bias_act.h -- internal launcher declarations (see bias_act.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

struct BiasActProblem {
    int io_dtype;                 // 0 = f32, 2 = bf16: x, skip, y
    int bias_dtype;               // 0 = f32, 2 = bf16 (bf16 only with a bf16 activation)
    int64_t rows;
    int cols;                     // multiple of 4 (f32) / 8 (bf16)
    int relu;
    float dropout_p;              // 0 = none
    uint64_t seed;
    const uint64_t *seed_dev;     // optional device word added to `seed`
};

bool bias_act_supported(int io_dtype, int bias_dtype, int cols);
// y may alias x (in place); bias and skip may be NULL
hipError_t bias_act_forward_launch(const BiasActProblem &p, const void *x, const void *bias, const void *skip, void *y,
                                   hipStream_t st);
// dx = y > 0 ? dy * scale : 0; dx may alias dy
hipError_t bias_act_backward_launch(int io_dtype, const void *dy, const void *y, void *dx, int64_t rows, int cols, float scale,
                                    hipStream_t st);

}  // namespace mdetr
