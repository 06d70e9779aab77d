// monodetr_amd/csrc/wfold.hip -- the frozen-BatchNorm fold of every TRAINABLE backbone convolution in one launch each way.
//
// The reference runs conv -> FrozenBatchNorm2d (lib/models/monodetr/backbone.py:27-64: y = x * scale + shift with
// scale = weight * rsqrt(running_var + eps)) as two passes over every activation; this repository folds the scale into the
// weight, conv(x, W * scale[:, None, None, None]) + shift (monodetr/backbone.py).  With trainable weights the fold is part of
// every iteration: folded = bf16(W * scale) forward, dW = float(dfolded) * scale backward -- 40 tensors, 23 M elements.  As
// multi-tensor framework calls that was two launches each way at a tenth of the HBM rate (full-size fp32 scale tensors read
// beside the weights), and every 3x3 convolution's input-gradient kernel copied its weight into [C][tap][N] order first
// (13 launches).  Here:
//   fold:   folded[o][t][c]  = bf16(w[o][t][c] * scale[o])        (OHWI: the channels-last parameter as it lies in memory)
//           foldedT[c][t][o] = the same value                     (3x3 only: the operand of the input-gradient kernels)
//   unfold: dw[o][t][c]      = float(dfolded[o][t][c]) * scale[o]
// Bit for bit what the framework calls computed (one fp32 product, one rounding).  Work unit of `fold`: a 32 (o) x 64 (c) tile
// of one tap, 256 threads, 8 values per thread; the transposed copy goes through LDS so that both stores are 16-byte pieces of
// whole 64 / 128-byte runs.  Tensor descriptors travel as kernel arguments (the folded tensors are new allocations every
// iteration; a captured graph bakes the addresses into its node).  HBM-bound: 4 + 2 (+ 2) bytes per element forward, 2 + 4 backward.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mdetr_wave.h>

#include "msda.h"       // profile scopes
#include "wfold.h"

namespace mdetr {
namespace {

constexpr int kFoldThreads = 256;
constexpr int kTileO = 32, kTileC = 64, kPitch = kTileC + 2;     // LDS rows of 66 bf16 = 33 dwords

struct FoldArgs {
    const float *w[kFoldTensors];
    const float *scale[kFoldTensors];
    __bf16 *folded[kFoldTensors];
    __bf16 *foldedT[kFoldTensors];
    int O[kFoldTensors], C[kFoldTensors], taps[kFoldTensors];
    int blk_begin[kFoldTensors + 1];
    int n;
};

struct UnfoldArgs {
    const __bf16 *g[kFoldTensors];
    const float *scale[kFoldTensors];
    float *dw[kFoldTensors];
    int row_len[kFoldTensors];                                   // taps * C: elements that share a scale
    int64_t numel[kFoldTensors];
    int blk_begin[kFoldTensors + 1];
    int n;
};

__device__ __forceinline__ int find_tensor(const int *blk_begin, int n, int b)
{
    int i = 0;
    while (i + 1 < n && b >= blk_begin[i + 1]) ++i;              // (uniform: a scalar loop over <= 48 entries)
    return i;
}

__global__ __launch_bounds__(kFoldThreads)
void fold_kernel(const FoldArgs a)
{
    __shared__ __bf16 tile[kTileO * kPitch];
    const int b = static_cast<int>(blockIdx.x);
    const int i = find_tensor(a.blk_begin, a.n, b);
    const int O = a.O[i], C = a.C[i], taps = a.taps[i];
    const int tiles_c = (C + kTileC - 1) / kTileC, tiles_o = (O + kTileO - 1) / kTileO;
    int r = b - a.blk_begin[i];
    const int tc = r % tiles_c; r /= tiles_c;
    const int to = r % tiles_o;
    const int t = r / tiles_o;
    const int tid = threadIdx.x, row = tid >> 3, piece = tid & 7;
    const int o = to * kTileO + row, c = tc * kTileC + piece * 8;
    const bool live = o < O && c < C;                            // C % 8 == 0
    bf16x8 v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = static_cast<__bf16>(0.f);
    if (live) {
        const int64_t off = (static_cast<int64_t>(o) * taps + t) * C + c;
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(a.w[i] + off), hi = *reinterpret_cast<const f32x4 *>(a.w[i] + off + 4);
        const float s = a.scale[i][o];
        v[0] = static_cast<__bf16>(lo.x * s); v[1] = static_cast<__bf16>(lo.y * s); v[2] = static_cast<__bf16>(lo.z * s); v[3] = static_cast<__bf16>(lo.w * s);
        v[4] = static_cast<__bf16>(hi.x * s); v[5] = static_cast<__bf16>(hi.y * s); v[6] = static_cast<__bf16>(hi.z * s); v[7] = static_cast<__bf16>(hi.w * s);
        *reinterpret_cast<bf16x8 *>(a.folded[i] + off) = v;
    }
    __bf16 *outT = a.foldedT[i];
    if (outT == nullptr) return;                                 // (uniform)
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[row * kPitch + piece * 8 + k] = v[k];
    __syncthreads();
    // thread -> (column cc, 8 consecutive rows): foldedT[c][t][o0 .. o0 + 7]
    const int cc = tid >> 2, q = tid & 3;
    const int c2 = tc * kTileC + cc, o2 = to * kTileO + q * 8;
    if (c2 < C && o2 < O) {                                      // O % 8 == 0
        bf16x8 u;
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = tile[(q * 8 + k) * kPitch + cc];
        *reinterpret_cast<bf16x8 *>(outT + (static_cast<int64_t>(c2) * taps + t) * O + o2) = u;
    }
}

__global__ __launch_bounds__(kFoldThreads)
void unfold_kernel(const UnfoldArgs a)
{
    const int b = static_cast<int>(blockIdx.x);
    const int i = find_tensor(a.blk_begin, a.n, b);
    const int64_t e = (static_cast<int64_t>(b - a.blk_begin[i]) * kFoldThreads + threadIdx.x) * 8;
    if (e >= a.numel[i]) return;                                 // numel % 8 == 0
    const bf16x8 v = *reinterpret_cast<const bf16x8 *>(a.g[i] + e);
    const float s = a.scale[i][e / a.row_len[i]];                // row_len % 8 == 0: the 8 values share their row
    f32x4 lo, hi;
    lo.x = static_cast<float>(v[0]) * s; lo.y = static_cast<float>(v[1]) * s; lo.z = static_cast<float>(v[2]) * s; lo.w = static_cast<float>(v[3]) * s;
    hi.x = static_cast<float>(v[4]) * s; hi.y = static_cast<float>(v[5]) * s; hi.z = static_cast<float>(v[6]) * s; hi.w = static_cast<float>(v[7]) * s;
    *reinterpret_cast<f32x4 *>(a.dw[i] + e) = lo;
    *reinterpret_cast<f32x4 *>(a.dw[i] + e + 4) = hi;
}

}  // namespace

bool fold_shape_supported(int O, int C, int taps)
{
    return O > 0 && C > 0 && taps > 0 && O % 8 == 0 && C % 8 == 0 && static_cast<int64_t>(O) * C * taps < (1ll << 31);
}

hipError_t fold_weights_launch(int n, const void *const *w, const void *const *scale, void *const *folded, void *const *foldedT,
                               const int *O, const int *C, const int *taps, hipStream_t st)
{
    for (int i0 = 0; i0 < n; i0 += kFoldTensors) {
        FoldArgs a;
        a.n = n - i0 < kFoldTensors ? n - i0 : kFoldTensors;
        int blk = 0;
        double bytes = 0.0;
        for (int j = 0; j < kFoldTensors; ++j) {
            const bool on = j < a.n;
            a.w[j] = on ? static_cast<const float *>(w[i0 + j]) : nullptr;
            a.scale[j] = on ? static_cast<const float *>(scale[i0 + j]) : nullptr;
            a.folded[j] = on ? static_cast<__bf16 *>(folded[i0 + j]) : nullptr;
            a.foldedT[j] = (on && foldedT) ? static_cast<__bf16 *>(foldedT[i0 + j]) : nullptr;
            a.O[j] = on ? O[i0 + j] : 0; a.C[j] = on ? C[i0 + j] : 0; a.taps[j] = on ? taps[i0 + j] : 0;
            a.blk_begin[j] = blk;
            if (on) {
                blk += ((a.O[j] + kTileO - 1) / kTileO) * ((a.C[j] + kTileC - 1) / kTileC) * a.taps[j];
                bytes += static_cast<double>(a.O[j]) * a.C[j] * a.taps[j] * (a.foldedT[j] ? 8.0 : 6.0);
            }
        }
        a.blk_begin[kFoldTensors] = blk;
        for (int j = a.n; j < kFoldTensors; ++j) a.blk_begin[j] = blk;
        if (blk == 0) continue;
        ProfileScope prof(20, blk, st, 0.0, bytes / 1e3);
        hipLaunchKernelGGL(fold_kernel, dim3(static_cast<unsigned>(blk)), dim3(kFoldThreads), 0, st, a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t unfold_grads_launch(int n, const void *const *g, const void *const *scale, void *const *dw, const int *O, const int *C, const int *taps,
                               hipStream_t st)
{
    for (int i0 = 0; i0 < n; i0 += kFoldTensors) {
        UnfoldArgs a;
        a.n = n - i0 < kFoldTensors ? n - i0 : kFoldTensors;
        int blk = 0;
        double bytes = 0.0;
        for (int j = 0; j < kFoldTensors; ++j) {
            const bool on = j < a.n;
            a.g[j] = on ? static_cast<const __bf16 *>(g[i0 + j]) : nullptr;
            a.scale[j] = on ? static_cast<const float *>(scale[i0 + j]) : nullptr;
            a.dw[j] = on ? static_cast<float *>(dw[i0 + j]) : nullptr;
            a.row_len[j] = on ? C[i0 + j] * taps[i0 + j] : 1;
            a.numel[j] = on ? static_cast<int64_t>(O[i0 + j]) * C[i0 + j] * taps[i0 + j] : 0;
            a.blk_begin[j] = blk;
            if (on) {
                blk += static_cast<int>((a.numel[j] / 8 + kFoldThreads - 1) / kFoldThreads);
                bytes += static_cast<double>(a.numel[j]) * 6.0;
            }
        }
        a.blk_begin[kFoldTensors] = blk;
        for (int j = a.n; j < kFoldTensors; ++j) a.blk_begin[j] = blk;
        if (blk == 0) continue;
        ProfileScope prof(20, blk, st, 0.0, bytes / 1e3);
        hipLaunchKernelGGL(unfold_kernel, dim3(static_cast<unsigned>(blk)), dim3(kFoldThreads), 0, st, a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace mdetr
