// monodetr_amd/csrc/add_ln.h -- internal launcher declarations (see add_ln.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

struct AddLnProblem {
    int io_dtype;                 // 0 = f32, 2 = bf16: a, b, y, s and their gradients
    int param_dtype;              // 0 = f32, 2 = bf16: gamma, beta
    int64_t rows;
    int cols;                     // 128, 256 or 512
    float eps, dropout_p;
    uint64_t seed;
    const uint64_t *seed_dev;     // optional device word added to `seed`
};

hipError_t add_ln_forward_launch(const AddLnProblem &p, const void *a, const void *b, const void *gamma, const void *beta,
                                 void *y, void *s, float *stats, hipStream_t st);
// partial: [add_ln_partial_rows(rows), 2 * cols] fp32 -- per-block sums of dy * xhat (first cols) and dy (last cols)
int64_t add_ln_partial_rows(int64_t rows);
hipError_t add_ln_backward_launch(const AddLnProblem &p, const void *dy, const void *s, const void *gamma, const float *stats,
                                  void *da, void *db, float *partial, hipStream_t st);

}  // namespace mdetr
