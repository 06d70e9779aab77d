// monodetr_amd/csrc/adamw.h -- internal launcher declaration (see adamw.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// param_dtype: 0 = f32 (master == param), 2 = bf16 (master = fp32 copy, param = rounded model copy)
hipError_t adamw_launch(int param_dtype, void *param, float *master, const void *grad, float *exp_avg,
                        float *exp_avg_sq, int64_t n, int64_t n_no_decay, float beta1, float beta2, float eps,
                        float weight_decay, float step_host, const float *step_dev, hipStream_t st,
                        const double *count_dev = nullptr, const double *lr_dev = nullptr, float lr_host = 0.f);

// the gradients stay where they are: `grads_host[i]` = base address of tensor i's gradient (same dtype and memory layout as the
// parameter); tables as for gather_flat_launch (decimate.h): byte offsets of the tensors in the flat buffers, their byte sizes,
// (tensor, byte start) per workgroup and the first workgroup of every tensor (host, ntensors + 1 entries)
hipError_t adamw_gathered_launch(int param_dtype, void *param, float *master, const void *const *grads_host, int ntensors,
                                 const int *tensor_block_begin_host, const int64_t *dst_off, const int64_t *nbytes, const int *blk_tensor,
                                 const int64_t *blk_start, int chunk_bytes, float *exp_avg, float *exp_avg_sq, int64_t n_no_decay,
                                 float beta1, float beta2, float eps, float weight_decay, float step_host, const float *step_dev,
                                 hipStream_t st, const double *count_dev = nullptr, const double *lr_dev = nullptr, float lr_host = 0.f);

}  // namespace mdetr
