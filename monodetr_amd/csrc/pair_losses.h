// monodetr_amd/csrc/pair_losses.h -- internal launcher declarations (see pair_losses.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pair_losses_math.h"

namespace mdetr {

// workspace: zero on first use, left zero by every forward launch (the finalising block clears it)
int64_t pair_losses_workspace_bytes(int L, int B);

hipError_t pair_losses_forward_launch(const PairLossDims &d, const PairLossIn &in, const int32_t *num,
                                      float num_boxes, const float *num_boxes_dev, float *out, float *comp,
                                      void *workspace, hipStream_t st);

hipError_t pair_losses_backward_launch(const PairLossDims &d, const PairLossIn &in, const float *grad_out,
                                       const float *comp, float num_boxes, const float *num_boxes_dev,
                                       float *g_logits, float *g_boxes, float *g_dims, float *g_depths,
                                       float *g_angles, hipStream_t st);

}  // namespace mdetr
