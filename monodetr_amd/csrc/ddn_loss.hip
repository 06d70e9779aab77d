// monodetr_amd/csrc/ddn_loss.hip -- MonoDETR's depth-map loss (DDNLoss) in one launch forward, one backward.
//
// The PyTorch formulation paints the target map (box rasterisation, index_reduce amin), bins it, runs a
// log-softmax focal loss and a foreground/background balancer: ~60 framework kernels forward and ~40
// backward on a 8 x 81 x 24 x 80 tensor.  Per pixel it is: scan the image's <= 50 boxes for the nearest
// covering object, one pass over the 81 logits for the softmax statistics, one for the focal terms
// (ddn_loss_math.h).  One thread per pixel; the forward ends in a block reduction and the same
// last-block finalisation as pair_losses.hip; the backward recomputes the pixel and writes its 81
// gradients.  Latency-bound (15 360 threads); the point is the launch count.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ddn_loss.h"

namespace mdetr {
namespace {

__global__ __launch_bounds__(256)
void ddn_fwd_kernel(const DdnDims d, const float *__restrict__ logits, const float *__restrict__ boxes,
                    const float *__restrict__ depth, const uint8_t *__restrict__ valid, float *__restrict__ out,
                    void *__restrict__ ws)
{
    __shared__ float red[4];
    __shared__ bool last;
    float *sum = static_cast<float *>(ws);
    unsigned *done = reinterpret_cast<unsigned *>(sum + 1);
    const int64_t n = static_cast<int64_t>(d.B) * d.H * d.W;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    float v = 0.f;
    if (i < n) {
        const int b = static_cast<int>(i / (d.H * d.W)), r = static_cast<int>(i - static_cast<int64_t>(b) * d.H * d.W);
        const int y = r / d.W, x = r - y * d.W;
        bool fg;
        const int t = ddn_target(d, boxes + static_cast<int64_t>(b) * d.K * 4, depth + static_cast<int64_t>(b) * d.K,
                                 valid + static_cast<int64_t>(b) * d.K, x, y, fg);
        v = ddn_pixel(d, logits + b * d.sb + y * d.sh + x * d.sw, t, 0.f, nullptr) * (fg ? d.fg_weight : d.bg_weight);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sum, red[0] + red[1] + red[2] + red[3]);
        __threadfence();
        last = atomicAdd(done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last || threadIdx.x != 0) return;
    __threadfence();
    out[0] = __hip_atomic_load(sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / static_cast<float>(n);
    *sum = 0.f;                                           // leave the workspace clean for the next launch
    *done = 0u;
}

__global__ __launch_bounds__(256)
void ddn_bwd_kernel(const DdnDims d, const float *__restrict__ logits, const float *__restrict__ boxes,
                    const float *__restrict__ depth, const uint8_t *__restrict__ valid,
                    const float *__restrict__ grad_out, float *__restrict__ grad_logits)
{
    const int64_t n = static_cast<int64_t>(d.B) * d.H * d.W;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const int b = static_cast<int>(i / (d.H * d.W)), r = static_cast<int>(i - static_cast<int64_t>(b) * d.H * d.W);
    const int y = r / d.W, x = r - y * d.W;
    bool fg;
    const int t = ddn_target(d, boxes + static_cast<int64_t>(b) * d.K * 4, depth + static_cast<int64_t>(b) * d.K,
                             valid + static_cast<int64_t>(b) * d.K, x, y, fg);
    const float scale = grad_out[0] * (fg ? d.fg_weight : d.bg_weight) / static_cast<float>(n);
    const int64_t off = b * d.sb + y * d.sh + x * d.sw;
    ddn_pixel(d, logits + off, t, scale, grad_logits + off);
}

}  // namespace

hipError_t ddn_loss_forward_launch(const DdnDims &d, const float *logits, const float *boxes, const float *depth,
                                   const uint8_t *valid, float *out, void *workspace, hipStream_t st)
{
    const int64_t n = static_cast<int64_t>(d.B) * d.H * d.W;
    hipLaunchKernelGGL(ddn_fwd_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st,
                       d, logits, boxes, depth, valid, out, workspace);
    return hipGetLastError();
}

hipError_t ddn_loss_backward_launch(const DdnDims &d, const float *logits, const float *boxes, const float *depth,
                                    const uint8_t *valid, const float *grad_out, float *grad_logits, hipStream_t st)
{
    const int64_t n = static_cast<int64_t>(d.B) * d.H * d.W;
    hipLaunchKernelGGL(ddn_bwd_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st,
                       d, logits, boxes, depth, valid, grad_out, grad_logits);
    return hipGetLastError();
}

}  // namespace mdetr
