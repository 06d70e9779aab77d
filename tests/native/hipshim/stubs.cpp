// tests/native/hipshim/stubs.cpp -- TEST INFRASTRUCTURE: launchers of the one kernel family the CPU emulation does not
// build (dense attention: GPU-validated, large, and its source keeps its own fragment types).  They let the unmodified
// capi.hip link; calling one reports hipErrorNotSupported through the C ABI's normal error path.
#include <hip/hip_runtime.h>

#include "attn.h"

namespace mdetr {

hipError_t attn_forward_launch(const AttnProblem &, void *, float *, hipStream_t) { return hipErrorNotSupported; }
hipError_t attn_backward_launch(const AttnProblem &, const void *, const void *, const float *, float *, void *, void *, void *, hipStream_t) { return hipErrorNotSupported; }

}  // namespace mdetr
