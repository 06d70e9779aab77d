// monodetr_amd/csrc/group_norm.h -- internal launcher declarations (see group_norm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

struct GroupNormProblem {
    int io_dtype;                 // 0 = f32, 2 = bf16: x, y, dy, dx
    int param_dtype;              // 0 = f32, 2 = bf16: gamma, beta, dgamma / dbeta
    int n;                        // images
    int64_t hw;                   // pixels per image
    int c, groups;                // channels (innermost, channels-last), groups: c / groups == 8
    float eps;
    int relu;                     // y = max(0, gn(x))
};

bool group_norm_supported(int io_dtype, int param_dtype, int c, int groups);
int64_t group_norm_workspace_bytes(int n, int64_t hw, int c, int groups);
// stats: [n, groups, 2] fp32 (mean, 1 / sqrt(var + eps)), kept for the backward
hipError_t group_norm_forward_launch(const GroupNormProblem &p, const void *x, const void *gamma, const void *beta, void *y,
                                     float *stats, void *workspace, hipStream_t st);
// dparams: [2, c] in param_dtype (dgamma row, dbeta row)
hipError_t group_norm_backward_launch(const GroupNormProblem &p, const void *dy, const void *x, const void *gamma, const void *beta,
                                      const float *stats, void *dx, void *dparams, void *workspace, hipStream_t st);

}  // namespace mdetr
