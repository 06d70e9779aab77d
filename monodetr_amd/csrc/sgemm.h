// monodetr_amd/csrc/sgemm.h -- internal launcher declarations (see sgemm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/monodetr_amd.h"

namespace mdetr {

// Validates a group (shapes, strides, alignment of the vectorised paths is decided per operand inside the kernel); returns a
// message or nullptr.
const char *sgemm_check(int mode, const mdetr_sgemm_problem *p, int nprob);
// bytes of scratch a group needs (TN groups whose contraction is cut into parts for occupancy; 0 otherwise)
int64_t sgemm_workspace_bytes(int mode, const mdetr_sgemm_problem *p, int nprob);
hipError_t sgemm_launch(int mode, const mdetr_sgemm_problem *p, int nprob, void *workspace, hipStream_t st);

}  // namespace mdetr
