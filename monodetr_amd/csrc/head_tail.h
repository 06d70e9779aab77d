// monodetr_amd/csrc/head_tail.h -- internal launcher declarations (see head_tail.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

struct HeadTailDims {
    int L, B, Q;          // decoder levels, images, queries
    int nd0;              // components of the initial reference (2 or 6)
    int H, W;             // the weighted depth map
};

// new_ref[T, 6] = sigmoid(delta[T, 6] + inverse_sigmoid(ref[T, nd]) on the first nd components)
hipError_t box_refine_launch(const float *delta, const float *ref, float *out, int64_t T, int nd, hipStream_t st);
hipError_t head_tail_forward_launch(const HeadTailDims &d, const float *delta, const float *init_ref, const float *inter_refs,
                                    const float *size3d, const float *depth_reg, const float *depth_map, const float *img_h,
                                    const float *focal, float *coord, float *depth_ave, hipStream_t st);
hipError_t head_tail_backward_launch(const HeadTailDims &d, const float *init_ref, const float *size3d, const float *depth_reg,
                                     const float *img_h, const float *focal, const float *coord, const float *g_coord,
                                     const float *g_depth, float *g_delta, float *g_init_ref, float *g_size3d, float *g_depth_reg,
                                     float *g_map, hipStream_t st);

}  // namespace mdetr
