// monodetr_amd/csrc/decimate.hip -- every second pixel of a channels-last activation, and the adjoint.
//
// The projection shortcut of a ResNet stage's first block is a 1x1 convolution of stride 2 (torchvision Bottleneck.downsample
// behind lib/models/monodetr/backbone.py:93-106; layer2 / 3 / 4: 256 -> 512, 512 -> 1024, 1024 -> 2048 channels).  A 1x1
// convolution reads only the pixels it keeps, so it IS a token GEMM over the decimated image:
//     y = W x[:, ::2, ::2]          dx[:, ::2, ::2] = W^T dy (zero elsewhere)          dW = dy^T x[:, ::2, ::2]
// The GEMMs go where the stride-1 1x1 convolutions go (hipBLASLt near the HBM rate, split-K weight gradient,
// monodetr/linear.py); what is left for this file is the gather and its adjoint, two HBM streams:
//   forward : reads the kept quarter of x, writes y                     -- bytes = 2 B OH OW C e
//   backward: reads dy, writes ALL of dx (the zeros are part of the result) -- bytes = B (OH OW + H W) C e
// 16 bytes per thread; consecutive threads cover a pixel's channels, then the next pixel of the row.
// (round 3's conv_taps.hip ran these shapes as one-tap implicit GEMMs at 105 - 117 us forward, 74 - 84 us input gradient,
// against 25 - 45 us for gather + GEMM.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "decimate.h"

namespace mdetr {
namespace {

struct DecDims {
    int B, H, W, OH, OW;
    int pieces;                 // 16-byte pieces per pixel
};

__global__ __launch_bounds__(256)
void decimate2_fwd_kernel(const uint4 *__restrict__ x, uint4 *__restrict__ y, const DecDims d)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t total = static_cast<int64_t>(d.B) * d.OH * d.OW * d.pieces;
    if (i >= total) return;
    const int piece = static_cast<int>(i % d.pieces);
    int64_t pix = i / d.pieces;
    const int ow = static_cast<int>(pix % d.OW); pix /= d.OW;
    const int oh = static_cast<int>(pix % d.OH);
    const int b = static_cast<int>(pix / d.OH);
    y[i] = x[((static_cast<int64_t>(b) * d.H + 2 * oh) * d.W + 2 * ow) * d.pieces + piece];
}

__global__ __launch_bounds__(256)
void decimate2_bwd_kernel(const uint4 *__restrict__ dy, uint4 *__restrict__ dx, const DecDims d)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t total = static_cast<int64_t>(d.B) * d.H * d.W * d.pieces;
    if (i >= total) return;
    const int piece = static_cast<int>(i % d.pieces);
    int64_t pix = i / d.pieces;
    const int w = static_cast<int>(pix % d.W); pix /= d.W;
    const int h = static_cast<int>(pix % d.H);
    const int b = static_cast<int>(pix / d.H);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (!(h & 1) && !(w & 1)) v = dy[((static_cast<int64_t>(b) * d.OH + (h >> 1)) * d.OW + (w >> 1)) * d.pieces + piece];
    dx[i] = v;
}

// ---- 3x3 / stride 2 / pad 1 max pooling of a channels-last bf16 activation (torchvision ResNet.maxpool behind backbone.py:93-106;
// the stem is frozen, so forward only).  8 channels (16 bytes) per thread, the nine window pixels read straight from global
// memory (neighbouring threads share them through the cache): reads x once, writes y -- the framework's NHWC kernel takes 80 us
// for the 126 MB -> 31 MB of the training shape, a quarter of the HBM rate.
__global__ __launch_bounds__(256)
void maxpool3x3s2_bf16_kernel(const uint4 *__restrict__ x, uint4 *__restrict__ y, const DecDims d)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t total = static_cast<int64_t>(d.B) * d.OH * d.OW * d.pieces;
    if (i >= total) return;
    const int piece = static_cast<int>(i % d.pieces);
    int64_t pix = i / d.pieces;
    const int ow = static_cast<int>(pix % d.OW); pix /= d.OW;
    const int oh = static_cast<int>(pix % d.OH);
    const int b = static_cast<int>(pix / d.OH);
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int h = 2 * oh + t - 1;
        if (h < 0 || h >= d.H) continue;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int w = 2 * ow + e - 1;
            if (w < 0 || w >= d.W) continue;
            const uint4 v = x[((static_cast<int64_t>(b) * d.H + h) * d.W + w) * d.pieces + piece];
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {                               // bf16 -> fp32 is a 16-bit shift; the maximum of bf16 values is one of them
                m[2 * k] = fmaxf(m[2 * k], __uint_as_float(u[k] << 16));
                m[2 * k + 1] = fmaxf(m[2 * k + 1], __uint_as_float(u[k] & 0xFFFF0000u));
            }
        }
    }
    uint4 o;
    o.x = (__float_as_uint(m[0]) >> 16) | (__float_as_uint(m[1]) & 0xFFFF0000u);
    o.y = (__float_as_uint(m[2]) >> 16) | (__float_as_uint(m[3]) & 0xFFFF0000u);
    o.z = (__float_as_uint(m[4]) >> 16) | (__float_as_uint(m[5]) & 0xFFFF0000u);
    o.w = (__float_as_uint(m[6]) >> 16) | (__float_as_uint(m[7]) & 0xFFFF0000u);
    y[i] = o;
}

// ---- gather of many dense tensors into one flat buffer (the optimizer's flat gradient buffer, helpers/optimizer_helper.FusedAdamW:
// the reference's AdamW walks the parameters one by one, optimizer_helper.py:69-129).  The framework's multi-tensor copy moves the
// 124 MB of gradients at 1.2 TB/s in six launches (0.21 ms per iteration); here a block owns one chunk of one tensor: table
// blk_tensor / blk_start (built once, on the device); the per-tensor SOURCE pointers travel as a kernel argument (autograd allocates
// gradients anew each iteration: no device table to refresh, nothing to race with, and a captured graph bakes them into its node),
// kGatherPtrs tensors per launch.  16-byte vectors when source and destination of the chunk allow, elements otherwise.
constexpr int kGatherPtrs = 256;
struct GatherPtrs { const unsigned char *p[kGatherPtrs]; };

__global__ __launch_bounds__(256)
void gather_flat_kernel(const GatherPtrs src, int tensor0, int block0, unsigned char *__restrict__ dst, const int64_t *__restrict__ dst_off,
                        const int64_t *__restrict__ nbytes, const int *__restrict__ blk_tensor, const int64_t *__restrict__ blk_start, int chunk_bytes)
{
    const int b = block0 + static_cast<int>(blockIdx.x);
    const int i = blk_tensor[b];
    const int64_t s0 = blk_start[b];
    const int64_t left = nbytes[i] - s0;
    const int cnt = static_cast<int>(left < chunk_bytes ? left : chunk_bytes);
    const unsigned char *sp = src.p[i - tensor0] + s0;
    unsigned char *dp = dst + dst_off[i] + s0;
    if (((reinterpret_cast<uintptr_t>(sp) | reinterpret_cast<uintptr_t>(dp)) & 15) == 0) {
        const int nv = cnt >> 4;
        for (int k = threadIdx.x; k < nv; k += 256) reinterpret_cast<uint4 *>(dp)[k] = reinterpret_cast<const uint4 *>(sp)[k];
        for (int k = (nv << 4) + threadIdx.x; k < cnt; k += 256) dp[k] = sp[k];
    } else if (((reinterpret_cast<uintptr_t>(sp) | reinterpret_cast<uintptr_t>(dp) | cnt) & 1) == 0) {
        for (int k = threadIdx.x; k < (cnt >> 1); k += 256) reinterpret_cast<unsigned short *>(dp)[k] = reinterpret_cast<const unsigned short *>(sp)[k];
    } else {
        for (int k = threadIdx.x; k < cnt; k += 256) dp[k] = sp[k];
    }
}

DecDims dims(int B, int H, int W, int64_t pixel_bytes)
{
    DecDims d;
    d.B = B; d.H = H; d.W = W; d.OH = (H + 1) / 2; d.OW = (W + 1) / 2;
    d.pieces = static_cast<int>(pixel_bytes / 16);
    return d;
}

}  // namespace

bool decimate2_supported(int64_t pixel_bytes, const void *x, const void *y)
{
    return pixel_bytes > 0 && pixel_bytes % 16 == 0 && pixel_bytes / 16 < (1 << 20) &&
           (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
}

hipError_t decimate2_forward_launch(const void *x, void *y, int B, int H, int W, int64_t pixel_bytes, hipStream_t st)
{
    const DecDims d = dims(B, H, W, pixel_bytes);
    const int64_t total = static_cast<int64_t>(B) * d.OH * d.OW * d.pieces;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(decimate2_fwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st,
                       static_cast<const uint4 *>(x), static_cast<uint4 *>(y), d);
    return hipGetLastError();
}

hipError_t decimate2_backward_launch(const void *dy, void *dx, int B, int H, int W, int64_t pixel_bytes, hipStream_t st)
{
    const DecDims d = dims(B, H, W, pixel_bytes);
    const int64_t total = static_cast<int64_t>(B) * H * W * d.pieces;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(decimate2_bwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st,
                       static_cast<const uint4 *>(dy), static_cast<uint4 *>(dx), d);
    return hipGetLastError();
}

hipError_t gather_flat_launch(const void *const *src_host, int ntensors, const int *tensor_block_begin_host, void *dst, const int64_t *dst_off,
                              const int64_t *nbytes, const int *blk_tensor, const int64_t *blk_start, int chunk_bytes, hipStream_t st)
{
    for (int t0 = 0; t0 < ntensors; t0 += kGatherPtrs) {
        const int t1 = t0 + kGatherPtrs < ntensors ? t0 + kGatherPtrs : ntensors;
        const int b0 = tensor_block_begin_host[t0], b1 = tensor_block_begin_host[t1];
        if (b1 <= b0) continue;
        GatherPtrs ptrs;
        for (int i = 0; i < kGatherPtrs; ++i) ptrs.p[i] = t0 + i < t1 ? static_cast<const unsigned char *>(src_host[t0 + i]) : nullptr;
        hipLaunchKernelGGL(gather_flat_kernel, dim3(static_cast<unsigned>(b1 - b0)), dim3(256), 0, st, ptrs, t0, b0,
                           static_cast<unsigned char *>(dst), dst_off, nbytes, blk_tensor, blk_start, chunk_bytes);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t maxpool3x3s2_bf16_launch(const void *x, void *y, int B, int H, int W, int C, hipStream_t st)
{
    DecDims d = dims(B, H, W, static_cast<int64_t>(C) * 2);
    d.OH = (H - 1) / 2 + 1; d.OW = (W - 1) / 2 + 1;                      // (H + 2 - 3) / 2 + 1
    const int64_t total = static_cast<int64_t>(B) * d.OH * d.OW * d.pieces;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(maxpool3x3s2_bf16_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st,
                       static_cast<const uint4 *>(x), static_cast<uint4 *>(y), d);
    return hipGetLastError();
}

}  // namespace mdetr
