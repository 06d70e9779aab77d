// monodetr_amd/csrc/msda.h -- internal launcher declarations (see msda.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// dtype: 0 = f32, 1 = f64 (MDETR_F32 / MDETR_F64 of include/monodetr_amd.h)
bool msda_fast_path(int dtype, int D, int L, int P);

hipError_t msda_forward_launch(int dtype, const void *value, const int64_t *shapes, const int64_t *lstart,
                               const void *loc, const void *attn, void *out,
                               int B, int S, int M, int D, int L, int Lq, int P, hipStream_t st);

hipError_t msda_backward_launch(int dtype, const void *value, const int64_t *shapes, const int64_t *lstart,
                                const void *loc, const void *attn, const void *grad_out,
                                void *grad_value, void *grad_loc, void *grad_attn,
                                int B, int S, int M, int D, int L, int Lq, int P, hipStream_t st);

// Extended backward: with host copies of the level geometry and a workspace, the self-attention
// (Lq == S) fp32 D == 32 case takes the tile-privatised grad_value path of msda_tiled.hip.
hipError_t msda_backward_launch_ex(int dtype, const void *value, const int64_t *shapes, const int64_t *lstart,
                                   const void *loc, const void *attn, const void *grad_out,
                                   void *grad_value, void *grad_loc, void *grad_attn,
                                   int B, int S, int M, int D, int L, int Lq, int P,
                                   const int64_t *shapes_host, const int64_t *lstart_host,
                                   void *workspace, int64_t workspace_bytes, hipStream_t st);

int64_t msda_tiled_workspace_bytes(const int64_t *shapes_h, const int64_t *start_h, int B, int S, int M, int D, int L, int Lq, int P);

hipError_t msda_tiled_grad_value_launch(const int64_t *shapes_h, const int64_t *start_h,
                                        const float *loc, const float *attn, const float *grad_out, float *grad_value,
                                        void *workspace, int64_t workspace_bytes,
                                        int B, int S, int M, int D, int L, int Lq, int P, bool absmax_ready, hipStream_t st);

hipError_t msda_indices_launch(int dtype, const int64_t *shapes, const void *loc, int32_t *idx,
                               int B, int M, int L, int Lq, int P, hipStream_t st);

// Zero `bytes` bytes at p with a store kernel.  Used instead of hipMemsetAsync: inside a captured
// hipGraph the memset NODE of a large, freshly mapped buffer was observed to leave stale data after
// the process mapped more device memory between replays (gradients of the first MSDA backward of a
// replay came back ~1e9); kernel nodes are not affected.
hipError_t zero_fill_launch(void *p, int64_t bytes, hipStream_t st);

// event-pair profiling of kernel launches (capi.hip owns the storage)
void profile_begin(int kind, int Lq, hipStream_t st);
void profile_end(hipStream_t st);

}  // namespace mdetr
