// monodetr_amd/csrc/wfold.h -- internal launcher declarations (see wfold.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

constexpr int kFoldTensors = 48;                                 // tensor descriptors per launch (kernel arguments)

// O, C multiples of 8 (16-byte pieces of both layouts)
bool fold_shape_supported(int O, int C, int taps);
// tensor i: w fp32 [O][taps][C] (a channels-last convolution weight as it lies in memory), scale fp32 [O] ->
// folded bf16 [O][taps][C] and, where foldedT[i] != NULL, foldedT bf16 [C][taps][O].  The pointer arrays live on the HOST.
hipError_t fold_weights_launch(int n, const void *const *w, const void *const *scale, void *const *folded, void *const *foldedT,
                               const int *O, const int *C, const int *taps, hipStream_t st);
// tensor i: dw fp32 [O][taps][C] = float(g bf16 [O][taps][C]) * scale[o]
hipError_t unfold_grads_launch(int n, const void *const *g, const void *const *scale, void *const *dw, const int *O, const int *C, const int *taps,
                               hipStream_t st);

}  // namespace mdetr
