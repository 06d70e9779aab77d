// monodetr_amd/csrc/adamw.hip -- the reference's AdamW step over one FLAT parameter group in one launch.
//
// The reference updates ~345 tensors in a Python loop with ~8 tiny kernels each
// (lib/helpers/optimizer_helper.py:78-129); the multi-tensor `_foreach` form still takes 86 launches
// and 1.7 ms of GPU time per step (profiles/r01h_bench_bf16_steady_kernel_stats.csv).  Here the
// parameters of a dtype live in one flat buffer (helpers/optimizer_helper.FusedAdamW), no-decay
// parameters first, so the whole update is a single HBM stream:
//   reads  g, m, v, p(master)   writes  m, v, p(master) [+ the bf16 model copy]
// = 28 bytes per fp32 parameter (30 with a bf16 copy): algorithmic bytes = 28-30 x n, HBM-bound.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "adamw.h"
#include "adamw_math.h"

namespace mdetr {
namespace {

template <typename G> __device__ __forceinline__ float to_f32(G x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<__hip_bfloat16>(__hip_bfloat16 x) { return __bfloat162float(x); }

__device__ __forceinline__ void load4(const float *p, float (&o)[4])
{
    const float4 v = *reinterpret_cast<const float4 *>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void load4(const __hip_bfloat16 *p, float (&o)[4])
{
    const uint2 v = *reinterpret_cast<const uint2 *>(p);             // bf16 -> fp32 is a 16-bit shift
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xFFFF0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xFFFF0000u);
}
__device__ __forceinline__ void store4(float *p, const float (&x)[4])
{
    *reinterpret_cast<float4 *>(p) = make_float4(x[0], x[1], x[2], x[3]);
}
__device__ __forceinline__ void store4(__hip_bfloat16 *p, const float (&x)[4])
{
    __hip_bfloat16 h[4] = {__float2bfloat16(x[0]), __float2bfloat16(x[1]), __float2bfloat16(x[2]), __float2bfloat16(x[3])};
    *reinterpret_cast<uint2 *>(p) = *reinterpret_cast<const uint2 *>(h);
}

// P = dtype of the model parameter / gradient (float or bf16); `master` is the fp32 copy the update
// runs on (== param when P is float).  All buffers are 16-byte aligned flat arrays; a thread owns 4
// consecutive elements per iteration (one 16-byte access per array).
template <typename P>
__global__ __launch_bounds__(256)
void adamw_kernel(P *__restrict__ param, float *__restrict__ master, const P *__restrict__ grad,
                  float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, int64_t n, int64_t n_no_decay,
                  AdamWCoef c, float step_host, const float *__restrict__ step_dev, const double *__restrict__ count_dev,
                  const double *__restrict__ lr_dev, float lr_host)
{
    // the bias-corrected step size: a host scalar, a device scalar, or -- count_dev -- computed here from the device-resident step
    // count and learning rate (a captured optimizer step: ten scalar launches per flat buffer otherwise)
    float step = step_dev ? *step_dev : step_host;
    if (count_dev) {
        const double t = *count_dev, lr = lr_dev ? *lr_dev : static_cast<double>(lr_host);
        step = static_cast<float>(lr * sqrt(1.0 - pow(static_cast<double>(c.beta2), t)) / (1.0 - pow(static_cast<double>(c.beta1), t)));
    }
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x * 4;
    for (int64_t i0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4; i0 < n; i0 += stride) {
        if (i0 + 4 <= n) {
            float g[4], m[4], v[4], p[4];
            load4(grad + i0, g); load4(exp_avg + i0, m); load4(exp_avg_sq + i0, v); load4(master + i0, p);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                p[j] = adamw_element(p[j], g[j], m[j], v[j], c, i0 + j < n_no_decay ? 0.f : c.weight_decay, step);
            store4(exp_avg + i0, m); store4(exp_avg_sq + i0, v); store4(master + i0, p);
            if (sizeof(P) == 2) store4(param + i0, p);               // rounded model copy of a bf16 parameter
        } else {
            for (int64_t i = i0; i < n; ++i) {
                float m = exp_avg[i], v = exp_avg_sq[i];
                const float q = adamw_element(master[i], to_f32<P>(grad[i]), m, v, c, i < n_no_decay ? 0.f : c.weight_decay, step);
                exp_avg[i] = m; exp_avg_sq[i] = v; master[i] = q;
                if (sizeof(P) == 2) param[i] = static_cast<P>(q);
            }
        }
    }
}

// The same update with the gradients where autograd left them: no flat gradient buffer and no copy into one (the multi-tensor copy
// of ~300 gradients was 7 launches and 0.12 ms per iteration, profiles/r06z7_opmap.txt).  A workgroup owns up to `chunk_bytes` of ONE
// parameter tensor (the tables of csrc/decimate.hip's gather_flat_kernel: tensor and byte offset per workgroup); the gradients' base
// addresses travel as kernel arguments, kAdamPtrs per launch (a captured graph bakes them into its node, like any other argument).
constexpr int kAdamPtrs = 256;
struct GradPtrs { const unsigned char *p[kAdamPtrs]; };

template <typename P>
__global__ __launch_bounds__(256)
void adamw_gathered_kernel(const GradPtrs grads, int tensor0, int block0, P *__restrict__ param, float *__restrict__ master,
                           float *__restrict__ exp_avg, float *__restrict__ exp_avg_sq, const int64_t *__restrict__ dst_off,
                           const int64_t *__restrict__ nbytes, const int *__restrict__ blk_tensor, const int64_t *__restrict__ blk_start,
                           int chunk_bytes, int64_t n_no_decay, AdamWCoef c, float step_host, const float *__restrict__ step_dev,
                           const double *__restrict__ count_dev, const double *__restrict__ lr_dev, float lr_host)
{
    float step = step_dev ? *step_dev : step_host;
    if (count_dev) {
        const double t = *count_dev, lr = lr_dev ? *lr_dev : static_cast<double>(lr_host);
        step = static_cast<float>(lr * sqrt(1.0 - pow(static_cast<double>(c.beta2), t)) / (1.0 - pow(static_cast<double>(c.beta1), t)));
    }
    const int b = block0 + static_cast<int>(blockIdx.x);
    const int ti = blk_tensor[b];
    const int64_t s0 = blk_start[b], left = nbytes[ti] - s0;
    const int ne = static_cast<int>((left < chunk_bytes ? left : chunk_bytes) / static_cast<int64_t>(sizeof(P)));     // elements of this workgroup
    const P *gp = reinterpret_cast<const P *>(grads.p[ti - tensor0] + s0);
    const int64_t e0 = (dst_off[ti] + s0) / static_cast<int64_t>(sizeof(P));                                        // first element in the flat arrays
    // the flat side is aligned by construction (tensors start at multiples of 64 elements, chunks at multiples of 16 bytes); a
    // gradient may be a slice of an exchange buffer at any element offset: its four elements then come as four loads
    const bool vec = (static_cast<uintptr_t>(e0 * sizeof(P)) & (4 * sizeof(P) - 1)) == 0;                            // (uniform)
    const bool gvec = (reinterpret_cast<uintptr_t>(gp) & (4 * sizeof(P) - 1)) == 0;                                  // (uniform)
    const int nv = vec ? ne & ~3 : 0;
    // two steps of the thread per iteration, the eight loads of both first (the second step of a ragged end re-reads the first's
    // elements and stores nothing)
    for (int ka = threadIdx.x * 4; ka < nv; ka += 2 * 256 * 4) {
        const int kb_ = ka + 256 * 4;
        const bool two = kb_ < nv;
        const int kb = two ? kb_ : ka;
        float g[2][4], m[2][4], v[2][4], q[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = u ? kb : ka;
            const int64_t i0 = e0 + k;
            if (gvec) {
                load4(gp + k, g[u]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) g[u][j] = to_f32<P>(gp[k + j]);
            }
            load4(exp_avg + i0, m[u]); load4(exp_avg_sq + i0, v[u]); load4(master + i0, q[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const int64_t i0 = e0 + (u ? kb : ka);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                q[u][j] = adamw_element(q[u][j], g[u][j], m[u][j], v[u][j], c, i0 + j < n_no_decay ? 0.f : c.weight_decay, step);
            store4(exp_avg + i0, m[u]); store4(exp_avg_sq + i0, v[u]); store4(master + i0, q[u]);
            if (sizeof(P) == 2) store4(param + i0, q[u]);
        }
    }
    for (int k = nv + threadIdx.x; k < ne; k += 256) {
        const int64_t i = e0 + k;
        float m = exp_avg[i], v = exp_avg_sq[i];
        const float q = adamw_element(master[i], to_f32<P>(gp[k]), m, v, c, i < n_no_decay ? 0.f : c.weight_decay, step);
        exp_avg[i] = m; exp_avg_sq[i] = v; master[i] = q;
        if (sizeof(P) == 2) param[i] = static_cast<P>(q);
    }
}

}  // namespace

hipError_t adamw_gathered_launch(int param_dtype, void *param, float *master, const void *const *grads_host, int ntensors,
                                 const int *tensor_block_begin_host, const int64_t *dst_off, const int64_t *nbytes, const int *blk_tensor,
                                 const int64_t *blk_start, int chunk_bytes, float *exp_avg, float *exp_avg_sq, int64_t n_no_decay,
                                 float beta1, float beta2, float eps, float weight_decay, float step_host, const float *step_dev,
                                 hipStream_t st, const double *count_dev, const double *lr_dev, float lr_host)
{
    const AdamWCoef c{beta1, beta2, eps, weight_decay};
    for (int t0 = 0; t0 < ntensors; t0 += kAdamPtrs) {
        const int t1 = t0 + kAdamPtrs < ntensors ? t0 + kAdamPtrs : ntensors;
        const int b0 = tensor_block_begin_host[t0], b1 = tensor_block_begin_host[t1];
        if (b1 <= b0) continue;
        GradPtrs ptrs;
        for (int i = 0; i < kAdamPtrs; ++i) ptrs.p[i] = t0 + i < t1 ? static_cast<const unsigned char *>(grads_host[t0 + i]) : nullptr;
        if (param_dtype == 2)
            hipLaunchKernelGGL(adamw_gathered_kernel<__hip_bfloat16>, dim3(static_cast<unsigned>(b1 - b0)), dim3(256), 0, st, ptrs, t0, b0,
                               static_cast<__hip_bfloat16 *>(param), master, exp_avg, exp_avg_sq, dst_off, nbytes, blk_tensor, blk_start,
                               chunk_bytes, n_no_decay, c, step_host, step_dev, count_dev, lr_dev, lr_host);
        else
            hipLaunchKernelGGL(adamw_gathered_kernel<float>, dim3(static_cast<unsigned>(b1 - b0)), dim3(256), 0, st, ptrs, t0, b0,
                               static_cast<float *>(param), master, exp_avg, exp_avg_sq, dst_off, nbytes, blk_tensor, blk_start,
                               chunk_bytes, n_no_decay, c, step_host, step_dev, count_dev, lr_dev, lr_host);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t adamw_launch(int param_dtype, void *param, float *master, const void *grad, float *exp_avg,
                        float *exp_avg_sq, int64_t n, int64_t n_no_decay, float beta1, float beta2, float eps,
                        float weight_decay, float step_host, const float *step_dev, hipStream_t st, const double *count_dev,
                        const double *lr_dev, float lr_host)
{
    if (n == 0) return hipSuccess;
    const AdamWCoef c{beta1, beta2, eps, weight_decay};
    int64_t blocks = (n + 256 * 4 - 1) / (256 * 4);
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (param_dtype == 2)
        hipLaunchKernelGGL(adamw_kernel<__hip_bfloat16>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st,
                           static_cast<__hip_bfloat16 *>(param), master, static_cast<const __hip_bfloat16 *>(grad),
                           exp_avg, exp_avg_sq, n, n_no_decay, c, step_host, step_dev, count_dev, lr_dev, lr_host);
    else
        hipLaunchKernelGGL(adamw_kernel<float>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st,
                           static_cast<float *>(param), master, static_cast<const float *>(grad),
                           exp_avg, exp_avg_sq, n, n_no_decay, c, step_host, step_dev, count_dev, lr_dev, lr_host);
    return hipGetLastError();
}

}  // namespace mdetr
