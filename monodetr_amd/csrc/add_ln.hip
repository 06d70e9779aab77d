// monodetr_amd/csrc/add_ln.hip -- y = LayerNorm(a + dropout(b)) in one pass each way (HBM bound).
//
// The model has ten such sites per forward (depthaware_transformer.py:331-353, :431-435, :456-510,
// depth_predictor/transformer.py:57-65), six of them on the 81 600 encoder tokens.  As framework operators a site
// costs, forward: dropout (read b; write b', mask) + add (read a, b'; write s) + LayerNorm (read s; write y, stats)
// = 7.5 row passes; backward: LayerNorm input gradient + two gamma / beta reduction kernels (read dy, s twice; write
// ds) + dropout backward (read ds, mask; write db) = 7.5 passes.  Here: forward reads a, b, writes y and s = a +
// dropout(b) (kept for the backward) -- 4 passes; backward reads dy, s, writes da (= ds) and db -- 4 passes -- with
// the dropout decision recomputed from a hash of the element index (no mask tensor, add_ln_math.h) and per-block
// partial sums of the gamma / beta gradients finished by the column-sum kernel.  Algorithmic bytes per site and
// direction = 4 * rows * cols * e (+ stats): 167 MB at 81 600 x 256 bf16.
//
// One wavefront per row: cols / 64 elements per lane in one 8- or 16-byte access (cols = 256: 4 elements),
// mean and variance by two wave reductions over registers (two-pass variance: no E[x^2] - E[x]^2 cancellation),
// fp32 arithmetic.  Four rows per 256-thread workgroup; the backward's workgroups walk a strided set of rows and
// keep the gamma / beta partial sums in registers until the end.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "add_ln.h"
#include "add_ln_math.h"

namespace mdetr {
namespace {

constexpr int kWavesLn = 4;
constexpr int kMaxBwdBlocks = 1024;

template <typename T> __device__ __forceinline__ float ld1(const T *p);
template <> __device__ __forceinline__ float ld1<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ld1<__hip_bfloat16>(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
template <typename T> __device__ __forceinline__ void st1(T *p, float v);
template <> __device__ __forceinline__ void st1<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<__hip_bfloat16>(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }

// a lane's E consecutive elements of a row in one aligned access (8 or 16 bytes; 2 x 16 for 8 floats)
template <typename T, int E> struct alignas(sizeof(T) * E > 16 ? 16 : sizeof(T) * E) RowVec { T v[E]; };

template <typename T, int E> __device__ __forceinline__ void load_row(const T *p, float (&v)[E])
{
    const RowVec<T, E> t = *reinterpret_cast<const RowVec<T, E> *>(p);
#pragma unroll
    for (int i = 0; i < E; ++i) v[i] = ld1<T>(&t.v[i]);
}
template <typename T, int E> __device__ __forceinline__ void store_row(T *p, const float (&v)[E])
{
    RowVec<T, E> t;
#pragma unroll
    for (int i = 0; i < E; ++i) st1<T>(&t.v[i], v[i]);
    *reinterpret_cast<RowVec<T, E> *>(p) = t;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename T, typename PT, int E>
__global__ __launch_bounds__(kWavesLn * 64)
void add_ln_fwd_kernel(const T *__restrict__ a, const T *__restrict__ b, const PT *__restrict__ gamma,
                       const PT *__restrict__ beta, T *__restrict__ y, T *__restrict__ s, float *__restrict__ stats,
                       int64_t rows, float eps, uint32_t thresh, float keep_scale, uint64_t seed,
                       const uint64_t *__restrict__ seed_dev)
{
    constexpr int C = 64 * E;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = static_cast<int64_t>(blockIdx.x) * kWavesLn + wave;
    if (row >= rows) return;                                    // whole wave: no exchange below is left short
    const uint64_t sd = seed + (seed_dev ? *seed_dev : 0ull);
    const int64_t at = row * C + lane * E;
    float v[E];
    load_row<T, E>(a + at, v);
    if (b) {
        float r[E];
        load_row<T, E>(b + at, r);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const bool keep = thresh == 0u || ln_hash(sd, static_cast<uint64_t>(at + i)) >= thresh;
            v[i] += keep ? r[i] * keep_scale : 0.f;
        }
        if (s) {                                                // round the saved sum exactly as the backward will read it
            store_row<T, E>(s + at, v);
#pragma unroll
            for (int i = 0; i < E; ++i) { T t; st1<T>(&t, v[i]); v[i] = ld1<T>(&t); }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) sum += v[i];
    const float mean = wave_sum(sum) * (1.0f / C);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) { const float d = v[i] - mean; sq += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * (1.0f / C) + eps);
    float o[E];
#pragma unroll
    for (int i = 0; i < E; ++i) o[i] = (v[i] - mean) * rstd * ld1<PT>(gamma + lane * E + i) + ld1<PT>(beta + lane * E + i);
    store_row<T, E>(y + at, o);
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

template <typename T, typename PT, int E>
__global__ __launch_bounds__(kWavesLn * 64)
void add_ln_bwd_kernel(const T *__restrict__ dy, const T *__restrict__ s, const PT *__restrict__ gamma,
                       const float *__restrict__ stats, T *__restrict__ da, T *__restrict__ db,
                       float *__restrict__ partial, int64_t rows, uint32_t thresh, float keep_scale, uint64_t seed,
                       const uint64_t *__restrict__ seed_dev)
{
    constexpr int C = 64 * E;
    __shared__ float red[kWavesLn][2 * C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t sd = seed + (seed_dev ? *seed_dev : 0ull);
    float g[E], acc_g[E], acc_b[E];
#pragma unroll
    for (int i = 0; i < E; ++i) { g[i] = ld1<PT>(gamma + lane * E + i); acc_g[i] = 0.f; acc_b[i] = 0.f; }
    for (int64_t row = static_cast<int64_t>(blockIdx.x) * kWavesLn + wave; row < rows; row += static_cast<int64_t>(gridDim.x) * kWavesLn) {
        const int64_t at = row * C + lane * E;
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        float d[E], x[E];
        load_row<T, E>(dy + at, d);
        load_row<T, E>(s + at, x);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            x[i] = (x[i] - mean) * rstd;                        // xhat
            acc_g[i] += d[i] * x[i];
            acc_b[i] += d[i];
            d[i] *= g[i];                                       // dy * gamma
            s1 += d[i];
            s2 += d[i] * x[i];
        }
        s1 = wave_sum(s1) * (1.0f / C);
        s2 = wave_sum(s2) * (1.0f / C);
        float o[E];
#pragma unroll
        for (int i = 0; i < E; ++i) o[i] = rstd * (d[i] - s1 - x[i] * s2);
        store_row<T, E>(da + at, o);
        if (db) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const bool keep = thresh == 0u || ln_hash(sd, static_cast<uint64_t>(at + i)) >= thresh;
                o[i] = keep ? o[i] * keep_scale : 0.f;
            }
            store_row<T, E>(db + at, o);
        }
    }
#pragma unroll
    for (int i = 0; i < E; ++i) { red[wave][lane * E + i] = acc_g[i]; red[wave][C + lane * E + i] = acc_b[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += kWavesLn * 64)
        partial[static_cast<int64_t>(blockIdx.x) * 2 * C + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

template <typename T, typename PT, int E>
hipError_t launch_fwd(const AddLnProblem &p, const void *a, const void *b, const void *gamma, const void *beta, void *y, void *s,
                      float *stats, hipStream_t st)
{
    const uint32_t thresh = b ? ln_threshold(p.dropout_p) : 0u;
    const float scale = p.dropout_p > 0.f ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
    hipLaunchKernelGGL((add_ln_fwd_kernel<T, PT, E>), dim3(static_cast<unsigned>((p.rows + kWavesLn - 1) / kWavesLn)), dim3(kWavesLn * 64), 0, st,
                       static_cast<const T *>(a), static_cast<const T *>(b), static_cast<const PT *>(gamma), static_cast<const PT *>(beta),
                       static_cast<T *>(y), static_cast<T *>(s), stats,
                       p.rows, p.eps, thresh, scale, p.seed, p.seed_dev);
    return hipGetLastError();
}

template <typename T, typename PT, int E>
hipError_t launch_bwd(const AddLnProblem &p, const void *dy, const void *s, const void *gamma, const float *stats, void *da, void *db,
                      float *partial, hipStream_t st)
{
    const uint32_t thresh = db ? ln_threshold(p.dropout_p) : 0u;
    const float scale = p.dropout_p > 0.f ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
    hipLaunchKernelGGL((add_ln_bwd_kernel<T, PT, E>), dim3(static_cast<unsigned>(add_ln_partial_rows(p.rows))), dim3(kWavesLn * 64), 0, st,
                       static_cast<const T *>(dy), static_cast<const T *>(s), static_cast<const PT *>(gamma), stats, static_cast<T *>(da),
                       static_cast<T *>(db), partial,
                       p.rows, thresh, scale, p.seed, p.seed_dev);
    return hipGetLastError();
}

}  // namespace

int64_t add_ln_partial_rows(int64_t rows)
{
    const int64_t blocks = (rows + kWavesLn - 1) / kWavesLn;
    return blocks < 1 ? 1 : (blocks > kMaxBwdBlocks ? kMaxBwdBlocks : blocks);
}

template <int E>
hipError_t dispatch_fwd(const AddLnProblem &p, const void *a, const void *b, const void *gamma, const void *beta, void *y, void *s,
                        float *stats, hipStream_t st)
{
    using BF = __hip_bfloat16;
    if (p.io_dtype == 2) return p.param_dtype == 2 ? launch_fwd<BF, BF, E>(p, a, b, gamma, beta, y, s, stats, st) : launch_fwd<BF, float, E>(p, a, b, gamma, beta, y, s, stats, st);
    return p.param_dtype == 2 ? launch_fwd<float, BF, E>(p, a, b, gamma, beta, y, s, stats, st) : launch_fwd<float, float, E>(p, a, b, gamma, beta, y, s, stats, st);
}

template <int E>
hipError_t dispatch_bwd(const AddLnProblem &p, const void *dy, const void *s, const void *gamma, const float *stats, void *da, void *db,
                        float *partial, hipStream_t st)
{
    using BF = __hip_bfloat16;
    if (p.io_dtype == 2) return p.param_dtype == 2 ? launch_bwd<BF, BF, E>(p, dy, s, gamma, stats, da, db, partial, st) : launch_bwd<BF, float, E>(p, dy, s, gamma, stats, da, db, partial, st);
    return p.param_dtype == 2 ? launch_bwd<float, BF, E>(p, dy, s, gamma, stats, da, db, partial, st) : launch_bwd<float, float, E>(p, dy, s, gamma, stats, da, db, partial, st);
}

hipError_t add_ln_forward_launch(const AddLnProblem &p, const void *a, const void *b, const void *gamma, const void *beta,
                                 void *y, void *s, float *stats, hipStream_t st)
{
    if (p.rows == 0) return hipSuccess;
    switch (p.cols) {
    case 128: return dispatch_fwd<2>(p, a, b, gamma, beta, y, s, stats, st);
    case 256: return dispatch_fwd<4>(p, a, b, gamma, beta, y, s, stats, st);
    case 512: return dispatch_fwd<8>(p, a, b, gamma, beta, y, s, stats, st);
    default: return hipErrorNotSupported;
    }
}

hipError_t add_ln_backward_launch(const AddLnProblem &p, const void *dy, const void *s, const void *gamma, const float *stats,
                                  void *da, void *db, float *partial, hipStream_t st)
{
    if (p.rows == 0) return hipSuccess;
    switch (p.cols) {
    case 128: return dispatch_bwd<2>(p, dy, s, gamma, stats, da, db, partial, st);
    case 256: return dispatch_bwd<4>(p, dy, s, gamma, stats, da, db, partial, st);
    case 512: return dispatch_bwd<8>(p, dy, s, gamma, stats, da, db, partial, st);
    default: return hipErrorNotSupported;
    }
}

}  // namespace mdetr
