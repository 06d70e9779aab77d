// monodetr_amd/csrc/ddn_loss.h -- internal launcher declarations (see ddn_loss.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_loss_math.h"

namespace mdetr {

// workspace: 16 bytes, zero on first use, left zero by every forward launch
hipError_t ddn_loss_forward_launch(const DdnDims &d, const float *logits, const float *boxes, const float *depth,
                                   const uint8_t *valid, float *out, void *workspace, hipStream_t st);
hipError_t ddn_loss_backward_launch(const DdnDims &d, const float *logits, const float *boxes, const float *depth,
                                    const uint8_t *valid, const float *grad_out, float *grad_logits, hipStream_t st);

}  // namespace mdetr
