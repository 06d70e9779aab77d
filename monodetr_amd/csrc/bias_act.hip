// monodetr_amd/csrc/bias_act.hip -- y = dropout(relu(x + bias[col] + skip)) in one pass, and its backward (HBM bound).
//
// The tails of the convolutions and token-wise linear layers of the model, which the reference runs as separate
// framework operators over the whole activation:
//   * ResNet bottleneck (lib/models/monodetr/backbone.py:100-102 -> torchvision Bottleneck.forward): 3x3 convolution
//     -> frozen BN -> ReLU, and 1x1 expansion -> frozen BN -> "+ identity" -> ReLU.  With the BN folded into the
//     convolution the library still adds the shift in its own kernel (MIOpen's OpTensor) and the ReLU and the residual
//     addition are two more passes: 3 (4) read+write passes over the activation where one suffices;
//   * FFN (depthaware_transformer.py:334-337, :431-435; depth_predictor/transformer.py:57-65): linear1 -> ReLU ->
//     Dropout: two passes (and a stored mask) after the GEMM.
// Channels-last activations make `col` the fastest index, so the tensor is a [rows, cols] matrix and the bias a
// vector over its columns.
//
// One 16-byte access per lane and tensor (8 bf16 / 4 fp32), four independent accesses in flight per lane,
// grid-stride over the vectors; when the grid stride is a multiple of the row's vector count (always for the model's
// power-of-two channel counts) a lane keeps its bias vector in registers.  fp32 arithmetic, one rounding at the
// store.  The dropout decision is the stateless hash of add_ln_math.h (element index, launch seed): no mask tensor;
// the backward needs no hash either, because y > 0 <=> (pre-activation > 0 and kept):  dx = y > 0 ? dy / (1 - p) : 0.
// Algorithmic bytes: forward (1 [+1 skip] reads + 1 write) * rows * cols * e; backward 2 reads + 1 write.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "add_ln_math.h"
#include "bias_act.h"

namespace mdetr {
namespace {

constexpr int kThreadsBa = 256;
constexpr int kMaxBlocksBa = 256 * 8;                        // 8 workgroups per CU

template <typename T> struct Io16;
template <> struct Io16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void unpack(const uint4 &v, float *f)
    {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    static __device__ __forceinline__ uint4 pack(const float *f)
    {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct Io16<__hip_bfloat16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void unpack(const uint4 &v, float *f)
    {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {                          // bf16 -> fp32 is a 16-bit shift
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
    static __device__ __forceinline__ unsigned rne(float f)    // round to nearest even; NaN stays NaN (quiet bit set)
    {
        const unsigned u = __float_as_uint(f);
        if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
        return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    }
    static __device__ __forceinline__ uint4 pack(const float *f)
    {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = rne(f[2 * i]) | (rne(f[2 * i + 1]) << 16);
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

struct BaParams {
    int64_t nvec;             // rows * cols / N
    int cvs;                  // vectors per row
    int relu;
    uint32_t thresh;          // dropout: keep iff hash >= thresh (0 = no dropout)
    float keep_scale;
    uint64_t seed;
    const uint64_t *seed_dev;
};

template <typename T, typename BT>
__device__ __forceinline__ void load_bias(const BT *__restrict__ bias, int cv, float *bv)
{
    constexpr int N = Io16<T>::N;
#pragma unroll
    for (int i = 0; i < N; ++i) bv[i] = 0.f;
    if (bias) {
        if constexpr (sizeof(BT) == 4) {
#pragma unroll
            for (int i = 0; i < N; i += 4) Io16<float>::unpack(*reinterpret_cast<const uint4 *>(bias + static_cast<int64_t>(cv) * N + i), bv + i);
        } else {
            static_assert(N == 8, "a 2-byte bias goes with a 2-byte activation");
            Io16<__hip_bfloat16>::unpack(*reinterpret_cast<const uint4 *>(bias + static_cast<int64_t>(cv) * N), bv);
        }
    }
}

template <typename T>
__device__ __forceinline__ uint4 ba_element(const uint4 &xv, const uint4 &sv, bool has_skip, const float *bv, int64_t vec,
                                            const BaParams &p, uint64_t sd)
{
    constexpr int N = Io16<T>::N;
    float f[N], s[N];
    Io16<T>::unpack(xv, f);
    if (has_skip) Io16<T>::unpack(sv, s);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float v = f[i] + bv[i];
        if (has_skip) v += s[i];
        if (p.relu) v = v < 0.f ? 0.f : v;                     // NaN passes through, as clamp_min does
        if (p.thresh != 0u) v = ln_hash(sd, static_cast<uint64_t>(vec * N + i)) >= p.thresh ? v * p.keep_scale : 0.f;
        f[i] = v;
    }
    return Io16<T>::pack(f);
}

// FIXED: the grid stride is a multiple of cvs, so a lane's column vector never changes.  SKIP: a residual operand
// (a template parameter: a run-time choice between a load and a constant becomes a select of ADDRESSES, i.e. flat
// loads from a scratch copy of the constant)
template <typename T, typename BT, bool FIXED, bool SKIP>
__global__ __launch_bounds__(kThreadsBa)
void bias_act_fwd_kernel(const T *x, const BT *__restrict__ bias, const T *skip, T *y, BaParams p)
{
    constexpr int N = Io16<T>::N;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreadsBa;
    int64_t v = static_cast<int64_t>(blockIdx.x) * kThreadsBa + threadIdx.x;
    const uint64_t sd = p.seed + (p.seed_dev ? *p.seed_dev : 0ull);
    const uint4 *xp = reinterpret_cast<const uint4 *>(x);
    const uint4 *sp = reinterpret_cast<const uint4 *>(skip);
    uint4 *yp = reinterpret_cast<uint4 *>(y);
    constexpr bool has_skip = SKIP;
    float bv[N];
    if (FIXED) load_bias<T, BT>(bias, static_cast<int>(v % p.cvs), bv);
    for (; v + 3 * stride < p.nvec; v += 4 * stride) {         // four independent 16-byte loads per tensor in flight
        uint4 a[4], s[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = xp[v + k * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (SKIP) s[k] = sp[v + k * stride]; else s[k] = make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!FIXED) load_bias<T, BT>(bias, static_cast<int>((v + k * stride) % p.cvs), bv);
            yp[v + k * stride] = ba_element<T>(a[k], s[k], has_skip, bv, v + k * stride, p, sd);
        }
    }
    for (; v < p.nvec; v += stride) {
        if (!FIXED) load_bias<T, BT>(bias, static_cast<int>(v % p.cvs), bv);
        const uint4 a = xp[v];
        uint4 s = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (SKIP) s = sp[v];
        yp[v] = ba_element<T>(a, s, has_skip, bv, v, p, sd);
    }
}

// dx = y > 0 ? dy * scale : 0   (scale = 1 / (1 - p) of the forward's dropout, 1 without)
template <typename T>
__global__ __launch_bounds__(kThreadsBa)
void bias_act_bwd_kernel(const T *dy, const T *__restrict__ y, T *dx, int64_t nvec, float scale)
{
    constexpr int N = Io16<T>::N;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kThreadsBa;
    int64_t v = static_cast<int64_t>(blockIdx.x) * kThreadsBa + threadIdx.x;
    const uint4 *gp = reinterpret_cast<const uint4 *>(dy);
    const uint4 *yp = reinterpret_cast<const uint4 *>(y);
    uint4 *dp = reinterpret_cast<uint4 *>(dx);
    auto one = [&](const uint4 &g, const uint4 &o) {
        float gf[N], of[N];
        Io16<T>::unpack(g, gf);
        Io16<T>::unpack(o, of);
#pragma unroll
        for (int i = 0; i < N; ++i) gf[i] = of[i] > 0.f ? gf[i] * scale : 0.f;
        return Io16<T>::pack(gf);
    };
    for (; v + 3 * stride < nvec; v += 4 * stride) {
        uint4 g[4], o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = gp[v + k * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = yp[v + k * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k) dp[v + k * stride] = one(g[k], o[k]);
    }
    for (; v < nvec; v += stride) dp[v] = one(gp[v], yp[v]);
}

int grid_for(int64_t nvec, int cvs)
{
    int64_t blocks = (nvec + kThreadsBa - 1) / kThreadsBa;
    if (blocks > kMaxBlocksBa) blocks = kMaxBlocksBa;
    if (blocks < 1) blocks = 1;
    // prefer a grid whose stride is a multiple of the row's vector count (register-resident bias): round down to a
    // multiple of cvs / gcd(cvs, 256) blocks when that keeps at least half of the grid
    int64_t g = cvs, t = kThreadsBa;
    while (t) { const int64_t r = g % t; g = t; t = r; }
    const int64_t unit = cvs / g;
    if (unit > 1 && blocks >= 2 * unit) blocks -= blocks % unit;
    return static_cast<int>(blocks);
}

template <typename T, typename BT>
hipError_t fwd_launch(const BiasActProblem &q, const void *x, const void *bias, const void *skip, void *y, hipStream_t st)
{
    constexpr int N = Io16<T>::N;
    BaParams p;
    p.cvs = q.cols / N;
    p.nvec = q.rows * p.cvs;
    p.relu = q.relu;
    p.thresh = q.dropout_p > 0.f ? ln_threshold(q.dropout_p) : 0u;
    p.keep_scale = q.dropout_p > 0.f ? 1.0f / (1.0f - q.dropout_p) : 1.0f;
    p.seed = q.seed;
    p.seed_dev = q.seed_dev;
    const int grid = grid_for(p.nvec, p.cvs);
    const bool fixed = (static_cast<int64_t>(grid) * kThreadsBa) % p.cvs == 0;
    const T *xs = static_cast<const T *>(x), *ss = static_cast<const T *>(skip);
    const BT *bs = static_cast<const BT *>(bias);
    T *ys = static_cast<T *>(y);
    const dim3 g(grid), b(kThreadsBa);
    if (fixed && skip)       hipLaunchKernelGGL((bias_act_fwd_kernel<T, BT, true, true>), g, b, 0, st, xs, bs, ss, ys, p);
    else if (fixed)          hipLaunchKernelGGL((bias_act_fwd_kernel<T, BT, true, false>), g, b, 0, st, xs, bs, ss, ys, p);
    else if (skip)           hipLaunchKernelGGL((bias_act_fwd_kernel<T, BT, false, true>), g, b, 0, st, xs, bs, ss, ys, p);
    else                     hipLaunchKernelGGL((bias_act_fwd_kernel<T, BT, false, false>), g, b, 0, st, xs, bs, ss, ys, p);
    return hipGetLastError();
}

}  // namespace

bool bias_act_supported(int io_dtype, int bias_dtype, int cols)
{
    if (io_dtype != 0 && io_dtype != 2) return false;
    if (bias_dtype != 0 && !(bias_dtype == 2 && io_dtype == 2)) return false;
    return cols > 0 && cols % (io_dtype == 2 ? 8 : 4) == 0;
}

hipError_t bias_act_forward_launch(const BiasActProblem &q, const void *x, const void *bias, const void *skip, void *y,
                                   hipStream_t st)
{
    if (q.rows == 0) return hipSuccess;
    if (q.io_dtype == 2)
        return q.bias_dtype == 2 ? fwd_launch<__hip_bfloat16, __hip_bfloat16>(q, x, bias, skip, y, st)
                                 : fwd_launch<__hip_bfloat16, float>(q, x, bias, skip, y, st);
    return fwd_launch<float, float>(q, x, bias, skip, y, st);
}

hipError_t bias_act_backward_launch(int io_dtype, const void *dy, const void *y, void *dx, int64_t rows, int cols, float scale,
                                    hipStream_t st)
{
    if (rows == 0) return hipSuccess;
    const int n = io_dtype == 2 ? 8 : 4;
    const int64_t nvec = rows * (cols / n);
    const int grid = grid_for(nvec, 1);
    if (io_dtype == 2)
        hipLaunchKernelGGL(bias_act_bwd_kernel<__hip_bfloat16>, dim3(grid), dim3(kThreadsBa), 0, st,
                           static_cast<const __hip_bfloat16 *>(dy), static_cast<const __hip_bfloat16 *>(y),
                           static_cast<__hip_bfloat16 *>(dx), nvec, scale);
    else
        hipLaunchKernelGGL(bias_act_bwd_kernel<float>, dim3(grid), dim3(kThreadsBa), 0, st, static_cast<const float *>(dy),
                           static_cast<const float *>(y), static_cast<float *>(dx), nvec, scale);
    return hipGetLastError();
}

}  // namespace mdetr
