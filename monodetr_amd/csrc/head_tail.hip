// monodetr_amd/csrc/head_tail.hip -- what follows the prediction heads, per query, in one launch each way.
//
// lib/models/monodetr/monodetr.py:226-253 turns the heads' raw outputs of every decoder level into the predictions:
//   box     = sigmoid(delta + inverse_sigmoid(reference))          reference = the initial points (level 0) / the previous level's box
//   h2d     = clamp((box_t + box_b) * image height, min = 1)
//   d_geo   = size3d_h / h2d * focal length
//   d_map   = bilinear(weighted depth map, 3-D centre (box_cx, box_cy), detached)      F.grid_sample, align_corners, zero padding
//   depth   = ((1 / (sigmoid(depth_reg_0) + 1e-6) - 1) + d_geo + d_map) / 3,   log-variance = depth_reg_1
// and depthaware_transformer.py:602-613 refines the reference between decoder layers (no gradient):
//   new_ref = sigmoid(delta + inverse_sigmoid(ref)).
// As framework operators that is ~130 launches per iteration on [3, 8, 550, <= 6] tensors (clamp, log, div, where, cat, stack,
// sigmoid, grid_sample and their backward nodes: 0.45 ms, every one of them a few microseconds of fill and drain).  Here: one thread
// per (level, image, query) forward and backward; the gradient of the depth map -- a scatter of 13 200 bilinear footprints into
// 8 x 24 x 80 cells -- is computed per CELL (each cell's thread walks its image's queries in order): deterministic, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "head_tail.h"

namespace mdetr {
namespace {

constexpr float kEps = 1e-5f;              // utils/misc.py inverse_sigmoid

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// inverse_sigmoid(x) = log(max(clamp(x, 0, 1), eps) / max(1 - clamp(x, 0, 1), eps))
__device__ __forceinline__ float inv_sigmoid(float x)
{
    x = fminf(fmaxf(x, 0.f), 1.f);
    return logf(fmaxf(x, kEps) / fmaxf(1.f - x, kEps));
}
// its derivative as autograd computes it: through the clamps (x outside [0, 1]: 0; a clamped branch: that term's share is 0)
__device__ __forceinline__ float inv_sigmoid_grad(float x)
{
    if (!(x >= 0.f && x <= 1.f)) return 0.f;
    const float a = x, b = 1.f - x;
    return (a >= kEps ? 1.f / a : 0.f) + (b >= kEps ? 1.f / b : 0.f);
}

__global__ __launch_bounds__(256)
void box_refine_kernel(const float *__restrict__ delta, const float *__restrict__ ref, float *__restrict__ out, int64_t T, int nd)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= T) return;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        float z = delta[i * 6 + c];
        if (c < nd) z += inv_sigmoid(ref[i * nd + c]);
        out[i * 6 + c] = sigmoidf_(z);
    }
}

struct Bilinear { int x0, y0; float wx, wy; };
// F.grid_sample(align_corners = True) of the point (cx, cy) in [0, 1]^2: pixel coordinates cx (W - 1), cy (H - 1)
__device__ __forceinline__ Bilinear footprint(float cx, float cy, int H, int W)
{
    // the reference goes through grid = (c - 0.5) * 2 and ((grid + 1) / 2) * (size - 1): the same arithmetic, in its order
    const float gx = (cx - 0.5f) * 2.f, gy = (cy - 0.5f) * 2.f;
    const float x = (gx + 1.f) * 0.5f * (W - 1), y = (gy + 1.f) * 0.5f * (H - 1);
    Bilinear f;
    const float fx = floorf(x), fy = floorf(y);
    f.x0 = static_cast<int>(fx); f.y0 = static_cast<int>(fy);
    f.wx = x - fx; f.wy = y - fy;
    return f;
}

__global__ __launch_bounds__(256)
void head_tail_fwd_kernel(const HeadTailDims d, const float *__restrict__ delta, const float *__restrict__ init_ref,
                          const float *__restrict__ inter_refs, const float *__restrict__ size3d, const float *__restrict__ depth_reg,
                          const float *__restrict__ depth_map, const float *__restrict__ img_h, const float *__restrict__ focal,
                          float *__restrict__ coord, float *__restrict__ depth_ave)
{
    const int i = blockIdx.x * 256 + threadIdx.x, BQ = d.B * d.Q;
    if (i >= d.L * BQ) return;
    const int l = i / BQ, bq = i - l * BQ, b = bq / d.Q;
    float c[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float z = delta[i * 6 + k];
        if (l == 0) { if (k < d.nd0) z += inv_sigmoid(init_ref[bq * d.nd0 + k]); }
        else z += inv_sigmoid(inter_refs[(static_cast<int64_t>(l - 1) * BQ + bq) * 6 + k]);
        c[k] = sigmoidf_(z);
        coord[i * 6 + k] = c[k];
    }
    const float h2d = fmaxf((c[4] + c[5]) * img_h[b], 1.f);
    const float geo = size3d[i * 3] / h2d * focal[b];
    const Bilinear f = footprint(c[0], c[1], d.H, d.W);
    const float *mp = depth_map + static_cast<int64_t>(b) * d.H * d.W;
    const auto at = [&](int y, int x) { return (x >= 0 && x < d.W && y >= 0 && y < d.H) ? mp[y * d.W + x] : 0.f; };
    const float m = at(f.y0, f.x0) * (1.f - f.wx) * (1.f - f.wy) + at(f.y0, f.x0 + 1) * f.wx * (1.f - f.wy)
                  + at(f.y0 + 1, f.x0) * (1.f - f.wx) * f.wy + at(f.y0 + 1, f.x0 + 1) * f.wx * f.wy;
    const float s = sigmoidf_(depth_reg[i * 2]);
    depth_ave[i * 2] = ((1.f / (s + 1e-6f) - 1.f) + geo + m) / 3.f;
    depth_ave[i * 2 + 1] = depth_reg[i * 2 + 1];
}

__global__ __launch_bounds__(256)
void head_tail_bwd_kernel(const HeadTailDims d, const float *__restrict__ init_ref, const float *__restrict__ size3d,
                          const float *__restrict__ depth_reg, const float *__restrict__ img_h, const float *__restrict__ focal,
                          const float *__restrict__ coord, const float *__restrict__ g_coord, const float *__restrict__ g_depth,
                          float *__restrict__ g_delta, float *__restrict__ g_init_ref, float *__restrict__ g_size3d,
                          float *__restrict__ g_depth_reg)
{
    const int i = blockIdx.x * 256 + threadIdx.x, BQ = d.B * d.Q;
    if (i >= d.L * BQ) return;
    const int l = i / BQ, bq = i - l * BQ, b = bq / d.Q;
    float c[6], gc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { c[k] = coord[i * 6 + k]; gc[k] = g_coord ? g_coord[i * 6 + k] : 0.f; }
    const float g0 = g_depth ? g_depth[i * 2] / 3.f : 0.f, g1 = g_depth ? g_depth[i * 2 + 1] : 0.f;
    const float raw = (c[4] + c[5]) * img_h[b], h2d = fmaxf(raw, 1.f), sz = size3d[i * 3], fo = focal[b];
    g_size3d[i * 3] = g0 * fo / h2d;
    g_size3d[i * 3 + 1] = 0.f;
    g_size3d[i * 3 + 2] = 0.f;
    if (raw >= 1.f) {                                             // (clamp(min = 1) passes the gradient where the input is >= the bound)
        const float gh = g0 * (-sz / (h2d * h2d)) * fo * img_h[b];
        gc[4] += gh; gc[5] += gh;
    }
    const float s = sigmoidf_(depth_reg[i * 2]), q = s + 1e-6f;
    g_depth_reg[i * 2] = g0 * (-1.f / (q * q)) * s * (1.f - s);
    g_depth_reg[i * 2 + 1] = g1;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float gz = gc[k] * c[k] * (1.f - c[k]);
        g_delta[i * 6 + k] = gz;
        if (l == 0 && k < d.nd0) g_init_ref[bq * d.nd0 + k] = gz * inv_sigmoid_grad(init_ref[bq * d.nd0 + k]);
    }
}

// d loss / d map[b, y, x] = sum over the image's (level, query) pairs of g_depth_0 / 3 * (the pair's bilinear weight on this cell).
// Deterministic, no atomics: 16 lanes share a cell -- lane i walks the pairs i, i + 16, ... of a batch in order, the 16 partial sums
// are added in lane order at the end -- and a workgroup owns 16 cells of one image; the image's pairs (footprint corner, fractions,
// gradient) are worked out once per workgroup into LDS in batches of 2 048.  (One thread per cell walking all 1 650 pairs, 64
// workgroups: 117 us, the longest kernel of the step after the two MSDA kernels -- a chain of dependent LDS reads.)
constexpr int kMapLanes = 16, kMapBatch = 2048;

__global__ __launch_bounds__(256)
void head_tail_map_grad_kernel(const HeadTailDims d, const float *__restrict__ coord, const float *__restrict__ g_depth, float *__restrict__ g_map)
{
    __shared__ short px[kMapBatch], py[kMapBatch];
    __shared__ float pwx[kMapBatch], pwy[kMapBatch], pg[kMapBatch];
    const int HW = d.H * d.W, cells_per_wg = 256 / kMapLanes, per_image = (HW + cells_per_wg - 1) / cells_per_wg;
    const int b = blockIdx.x / per_image, lane = threadIdx.x % kMapLanes;
    const int cell = (blockIdx.x - b * per_image) * cells_per_wg + threadIdx.x / kMapLanes;
    const int y = cell / d.W, x = cell - y * d.W;
    const int pairs = d.L * d.Q;
    float acc = 0.f;
    for (int p0 = 0; g_depth && p0 < pairs; p0 += kMapBatch) {
        const int n = pairs - p0 < kMapBatch ? pairs - p0 : kMapBatch;
        __syncthreads();
        for (int k = threadIdx.x; k < n; k += 256) {
            const int p = p0 + k, l = p / d.Q, q = p - l * d.Q, j = (l * d.B + b) * d.Q + q;
            const Bilinear f = footprint(coord[j * 6], coord[j * 6 + 1], d.H, d.W);
            px[k] = static_cast<short>(f.x0 < -2 ? -2 : (f.x0 > 32000 ? 32000 : f.x0)); py[k] = static_cast<short>(f.y0 < -2 ? -2 : (f.y0 > 32000 ? 32000 : f.y0)); pwx[k] = f.wx; pwy[k] = f.wy;
            pg[k] = g_depth[j * 2] / 3.f;
        }
        __syncthreads();
        for (int k = lane; k < n; k += kMapLanes) {
            const int dx = x - px[k], dy = y - py[k];
            if (dx < 0 || dx > 1 || dy < 0 || dy > 1) continue;
            acc += pg[k] * (dx ? pwx[k] : 1.f - pwx[k]) * (dy ? pwy[k] : 1.f - pwy[k]);
        }
    }
    // the 16 lanes of the cell, in lane order (the same on every run)
    float sum = 0.f;
    const int base = (threadIdx.x & 63) & ~(kMapLanes - 1);
#pragma unroll
    for (int i = 0; i < kMapLanes; ++i) sum += __shfl(acc, base + i);
    if (lane == 0 && cell < HW) g_map[b * HW + cell] = sum;
}

}  // namespace

hipError_t box_refine_launch(const float *delta, const float *ref, float *out, int64_t T, int nd, hipStream_t st)
{
    if (T == 0) return hipSuccess;
    hipLaunchKernelGGL(box_refine_kernel, dim3(static_cast<unsigned>((T + 255) / 256)), dim3(256), 0, st, delta, ref, out, T, nd);
    return hipGetLastError();
}

hipError_t head_tail_forward_launch(const HeadTailDims &d, const float *delta, const float *init_ref, const float *inter_refs,
                                    const float *size3d, const float *depth_reg, const float *depth_map, const float *img_h,
                                    const float *focal, float *coord, float *depth_ave, hipStream_t st)
{
    const int n = d.L * d.B * d.Q;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(head_tail_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d, delta, init_ref, inter_refs, size3d, depth_reg,
                       depth_map, img_h, focal, coord, depth_ave);
    return hipGetLastError();
}

hipError_t head_tail_backward_launch(const HeadTailDims &d, const float *init_ref, const float *size3d, const float *depth_reg,
                                     const float *img_h, const float *focal, const float *coord, const float *g_coord,
                                     const float *g_depth, float *g_delta, float *g_init_ref, float *g_size3d, float *g_depth_reg,
                                     float *g_map, hipStream_t st)
{
    const int n = d.L * d.B * d.Q;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(head_tail_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d, init_ref, size3d, depth_reg, img_h, focal, coord,
                       g_coord, g_depth, g_delta, g_init_ref, g_size3d, g_depth_reg);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (g_map) {
        const int blocks = d.B * ((d.H * d.W + 256 / kMapLanes - 1) / (256 / kMapLanes));
        hipLaunchKernelGGL(head_tail_map_grad_kernel, dim3(blocks), dim3(256), 0, st, d, coord, g_depth, g_map);      // (no g_depth: zeros)
        e = hipGetLastError();
    }
    return e;
}

}  // namespace mdetr
