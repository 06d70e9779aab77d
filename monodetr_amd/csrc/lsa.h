// monodetr_amd/csrc/lsa.h -- batched assignment solver launcher (lsa.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// layers * images * groups independent problems; problem (l, b, g) matches the num_targets[b] (<= kmax <= n)
// targets of image b to the n queries [g*n, (g+1)*n) using cost[(l*images + b) * img_stride + q * q_stride + t * t_stride].
hipError_t lsa_launch(const float *cost, const int *num_targets, int *assign, int layers, int images, int groups,
                      int n, int kmax, int64_t img_stride, int64_t q_stride, int64_t t_stride, hipStream_t st);

// same problems, with the matching cost evaluated in the kernel (matcher.py:55-84) from level-stacked
// predictions logits [layers*images, groups*n, num_classes], boxes [.., 6] and the padded ground truth
// labels [images, kmax] (int64), boxes3d [images, kmax, 6]
hipError_t lsa_fused_launch(const float *logits, const float *boxes, const int64_t *labels, const float *boxes3d,
                            const int *num_targets, int *assign, int layers, int images, int groups, int n, int kmax,
                            int num_classes, float w_class, float w_bbox, float w_center, float w_giou, float alpha,
                            hipStream_t st);

}  // namespace mdetr
