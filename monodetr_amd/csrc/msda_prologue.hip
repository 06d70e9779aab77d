// monodetr_amd/csrc/msda_prologue.hip -- softmax over a head's L*P samples + sampling-location arithmetic of
// MSDeformAttn.forward in one launch (and their gradients in one more), straight from the two projection
// outputs to the fp32 `sampling_locations` / `attention_weights` the sampling operator consumes.
//
// The module does this with a softmax, a divide, an add and -- in a bf16 model -- casts on [B, Lq, M, L, P(, 2)]
// tensors (10-20 M elements at the encoder shape): 5 launches forward and 5-7 backward per MSDA call, and in
// bf16 the locations are rounded to 8 bits of mantissa on the way.  Here one thread owns one (image, query,
// head): its 2 LP offsets and LP logits are contiguous, consecutive threads are consecutive heads of a query
// (coalesced), everything is evaluated in fp32 (msda_prologue_math.h).  HBM-bound: algorithmic bytes =
// B Lq M LP (3 e_io + 12) forward.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include <initializer_list>
#include <type_traits>

#include "msda_prologue.h"
#include "msda_prologue_math.h"

namespace mdetr {
namespace {

template <typename T> __device__ __forceinline__ float ldf(const T *p);
template <> __device__ __forceinline__ float ldf<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ldf<__hip_bfloat16>(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
template <typename T> __device__ __forceinline__ void stf(T *p, float v);
template <> __device__ __forceinline__ void stf<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<__hip_bfloat16>(__hip_bfloat16 *p, float v) { *p = __float2bfloat16(v); }

// TLP = L * P known at compile time (16 for MonoDETR: the per-sample arrays stay in registers) or 0 (any LP <= 64)
template <typename T, typename RT, int TLP>
__global__ __launch_bounds__(256)
void prologue_fwd_kernel(const PrologueDims d, const T *__restrict__ offsets, const T *__restrict__ logits,
                         const RT *__restrict__ ref, const int64_t *__restrict__ shapes, float *__restrict__ loc,
                         float *__restrict__ attn)
{
    const int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;              // (b, q, m)
    const int64_t total = static_cast<int64_t>(d.B) * d.Lq * d.M;
    if (u >= total) return;
    const int64_t bq = u / d.M;
    const int b = static_cast<int>(bq / d.Lq), q = static_cast<int>(bq - static_cast<int64_t>(b) * d.Lq);
    constexpr int CAP = TLP ? TLP : kPrologueMaxLP;
    const int LP = TLP ? TLP : d.L * d.P;
    float lg[CAP], at[CAP];
#pragma unroll
    for (int i = 0; i < (TLP ? TLP : 1); ++i) if (TLP) lg[i] = ldf<T>(logits + u * LP + i);
    if (!TLP) for (int i = 0; i < LP; ++i) lg[i] = ldf<T>(logits + u * LP + i);
    pro_softmax(lg, LP, at);
#pragma unroll
    for (int i = 0; i < (TLP ? TLP : 1); ++i) if (TLP) attn[u * LP + i] = at[i];
    if (!TLP) for (int i = 0; i < LP; ++i) attn[u * LP + i] = at[i];
    for (int l = 0; l < d.L; ++l) {
        float rl[6];
        const RT *rp = ref + b * d.rsb + q * d.rsq + l * d.rsl;
        for (int r = 0; r < d.R; ++r) rl[r] = ldf<RT>(rp + r);
        const float wh[2] = {static_cast<float>(shapes[2 * l + 1]), static_cast<float>(shapes[2 * l])};   // (W_l, H_l)
        for (int p = 0; p < d.P; ++p)
            for (int c = 0; c < 2; ++c) {
                const int64_t i = (u * LP + l * d.P + p) * 2 + c;
                loc[i] = pro_location(ldf<T>(offsets + i), rl, d.R, c, wh[c], d.P);
            }
    }
}

template <typename T, typename RT, int TLP>
__global__ __launch_bounds__(256)
void prologue_bwd_kernel(const PrologueDims d, const T *__restrict__ offsets, const RT *__restrict__ ref,
                         const int64_t *__restrict__ shapes, const float *__restrict__ attn,
                         const float *__restrict__ g_loc, const float *__restrict__ g_attn, T *__restrict__ g_offsets,
                         T *__restrict__ g_logits, float *__restrict__ g_ref)
{
    const int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t total = static_cast<int64_t>(d.B) * d.Lq * d.M;
    if (u >= total) return;
    const int64_t bq = u / d.M;
    const int b = static_cast<int>(bq / d.Lq), q = static_cast<int>(bq - static_cast<int64_t>(b) * d.Lq);
    constexpr int CAP = TLP ? TLP : kPrologueMaxLP;
    const int LP = TLP ? TLP : d.L * d.P;
    float at[CAP], ga[CAP], gl[CAP];
#pragma unroll
    for (int i = 0; i < (TLP ? TLP : 1); ++i) if (TLP) { at[i] = attn[u * LP + i]; ga[i] = g_attn[u * LP + i]; }
    if (!TLP) for (int i = 0; i < LP; ++i) { at[i] = attn[u * LP + i]; ga[i] = g_attn[u * LP + i]; }
    pro_softmax_backward(at, ga, LP, gl);
#pragma unroll
    for (int i = 0; i < (TLP ? TLP : 1); ++i) if (TLP) stf<T>(g_logits + u * LP + i, gl[i]);
    if (!TLP) for (int i = 0; i < LP; ++i) stf<T>(g_logits + u * LP + i, gl[i]);
    for (int l = 0; l < d.L; ++l) {
        float rl[6], gr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const RT *rp = ref + b * d.rsb + q * d.rsq + l * d.rsl;
        for (int r = 0; r < d.R; ++r) rl[r] = ldf<RT>(rp + r);
        const float wh[2] = {static_cast<float>(shapes[2 * l + 1]), static_cast<float>(shapes[2 * l])};
        for (int p = 0; p < d.P; ++p)
            for (int c = 0; c < 2; ++c) {
                const int64_t i = (u * LP + l * d.P + p) * 2 + c;
                const float off = d.R == 2 ? 0.f : ldf<T>(offsets + i);
                stf<T>(g_offsets + i, pro_location_backward(g_loc[i], off, rl, d.R, c, wh[c], d.P, g_ref ? gr : nullptr));
            }
        if (g_ref) {
            float *out = g_ref + ((static_cast<int64_t>(b) * d.Lq + q) * d.L + l) * d.R;
            for (int r = 0; r < d.R; ++r) unsafeAtomicAdd(out + r, gr[r]);                 // summed over the M heads
        }
    }
}

// ---- L = P = 4 (the model's configuration), 16-byte aligned tensors: the unit's 16 logits / 32 offsets / 16 weights /
// ---- 32 locations move as 16-byte lane accesses (2-8 per tensor instead of 16-32 scalar ones, each of which touches 64
// ---- different cache lines per wave), every array index is a compile-time constant (registers, no scratch)
template <typename T, int N> __device__ __forceinline__ void load_vec(const T *p, float (&v)[N])
{
    constexpr int E = 16 / sizeof(T);                              // elements per 16-byte access
    static_assert(N % E == 0, "whole 16-byte accesses");
    struct alignas(16) Pack { T e[E]; };
#pragma unroll
    for (int k = 0; k < N / E; ++k) {
        const Pack t = *reinterpret_cast<const Pack *>(p + k * E);
#pragma unroll
        for (int i = 0; i < E; ++i) v[k * E + i] = ldf<T>(&t.e[i]);
    }
}
template <typename T, int N> __device__ __forceinline__ void store_vec(T *p, const float (&v)[N])
{
    constexpr int E = 16 / sizeof(T);
    static_assert(N % E == 0, "whole 16-byte accesses");
    struct alignas(16) Pack { T e[E]; };
#pragma unroll
    for (int k = 0; k < N / E; ++k) {
        Pack t;
#pragma unroll
        for (int i = 0; i < E; ++i) stf<T>(&t.e[i], v[k * E + i]);
        *reinterpret_cast<Pack *>(p + k * E) = t;
    }
}

template <typename RT> __device__ __forceinline__ void load_ref(const RT *rp, int R, float (&rl)[6])
{
#pragma unroll
    for (int r = 0; r < 6; ++r) rl[r] = r < R ? ldf<RT>(rp + r) : 0.f;
}

template <typename T, typename RT>
__global__ __launch_bounds__(256)
void prologue_fwd_vec44(const PrologueDims d, const T *__restrict__ offsets, const T *__restrict__ logits,
                        const RT *__restrict__ ref, const int64_t *__restrict__ shapes, float *__restrict__ loc,
                        float *__restrict__ attn)
{
    const int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;              // (b, q, m)
    if (u >= static_cast<int64_t>(d.B) * d.Lq * d.M) return;
    const int64_t bq = u / d.M;
    const int b = static_cast<int>(bq / d.Lq), q = static_cast<int>(bq - static_cast<int64_t>(b) * d.Lq);
    float lg[16], at[16], off[32], lc[32];
    const int mh = static_cast<int>(u - bq * d.M);                                         // head
    load_vec<T, 16>(logits + (d.pl ? bq * d.pl + mh * 16 : u * 16), lg);
    load_vec<T, 32>(offsets + (d.po ? bq * d.po + mh * 32 : u * 32), off);
    pro_softmax(lg, 16, at);
    store_vec<float, 16>(attn + u * 16, at);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        float rl[6];
        load_ref<RT>(ref + b * d.rsb + q * d.rsq + l * d.rsl, d.R, rl);
        const float wh[2] = {static_cast<float>(shapes[2 * l + 1]), static_cast<float>(shapes[2 * l])};   // (W_l, H_l)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < 2; ++c) lc[(l * 4 + p) * 2 + c] = pro_location(off[(l * 4 + p) * 2 + c], rl, d.R, c, wh[c], 4);
    }
    store_vec<float, 32>(loc + u * 32, lc);
}

template <typename T, typename RT>
__global__ __launch_bounds__(256)
void prologue_bwd_vec44(const PrologueDims d, const T *__restrict__ offsets, const RT *__restrict__ ref,
                        const int64_t *__restrict__ shapes, const float *__restrict__ attn,
                        const float *__restrict__ g_loc, const float *__restrict__ g_attn, T *__restrict__ g_offsets,
                        T *__restrict__ g_logits, float *__restrict__ g_ref)
{
    const int64_t u = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (u >= static_cast<int64_t>(d.B) * d.Lq * d.M) return;
    const int64_t bq = u / d.M;
    const int b = static_cast<int>(bq / d.Lq), q = static_cast<int>(bq - static_cast<int64_t>(b) * d.Lq);
    float at[16], ga[16], gl[16], glc[32], off[32], go[32];
    load_vec<float, 16>(attn + u * 16, at);
    load_vec<float, 16>(g_attn + u * 16, ga);
    load_vec<float, 32>(g_loc + u * 32, glc);
    const int mh = static_cast<int>(u - bq * d.M);                                         // head
    if (d.R != 2) load_vec<T, 32>(offsets + (d.po ? bq * d.po + mh * 32 : u * 32), off);   // the 2-component form does not read the offsets
    else {
#pragma unroll
        for (int i = 0; i < 32; ++i) off[i] = 0.f;
    }
    pro_softmax_backward(at, ga, 16, gl);
    store_vec<T, 16>(g_logits + (d.pl ? bq * d.pl + mh * 16 : u * 16), gl);
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        float rl[6], gr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        load_ref<RT>(ref + b * d.rsb + q * d.rsq + l * d.rsl, d.R, rl);
        const float wh[2] = {static_cast<float>(shapes[2 * l + 1]), static_cast<float>(shapes[2 * l])};
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int i = (l * 4 + p) * 2 + c;
                go[i] = pro_location_backward(glc[i], off[i], rl, d.R, c, wh[c], 4, gr);    // (a select with nullptr would put gr in scratch)
            }
        if (g_ref) {
            float *out = g_ref + ((static_cast<int64_t>(b) * d.Lq + q) * d.L + l) * d.R;
#pragma unroll
            for (int r = 0; r < 6; ++r)
                if (r < d.R) unsafeAtomicAdd(out + r, gr[r]);                             // summed over the M heads
        }
    }
    store_vec<T, 32>(g_offsets + (d.po ? bq * d.po + mh * 32 : u * 32), go);
}

bool aligned16_all(std::initializer_list<const void *> ps)
{
    for (const void *p : ps)
        if (reinterpret_cast<uintptr_t>(p) & 15u) return false;
    return true;
}

}  // namespace

// reference points may stay fp32 while the projections' outputs are bf16 (a bf16 model body keeps its coordinates in fp32)
template <typename T, typename RT>
hipError_t launch_fwd(const PrologueDims &d, const void *offsets, const void *logits, const void *ref, const int64_t *shapes,
                      float *loc, float *attn, dim3 grid, hipStream_t st)
{
    const bool pitched = d.po != 0 || d.pl != 0;
    if (pitched && !(d.L == 4 && d.P == 4 && aligned16_all({offsets, logits, loc, attn}) && (d.po * sizeof(T)) % 16 == 0 && (d.pl * sizeof(T)) % 16 == 0))
        return hipErrorNotSupported;                                 // a packed projection output: the 16-byte form only
    if (d.L == 4 && d.P == 4 && aligned16_all({offsets, logits, loc, attn}))
        hipLaunchKernelGGL((prologue_fwd_vec44<T, RT>), grid, dim3(256), 0, st, d, static_cast<const T *>(offsets),
                           static_cast<const T *>(logits), static_cast<const RT *>(ref), shapes, loc, attn);
    else if (d.L * d.P == 16)
        hipLaunchKernelGGL((prologue_fwd_kernel<T, RT, 16>), grid, dim3(256), 0, st, d, static_cast<const T *>(offsets),
                           static_cast<const T *>(logits), static_cast<const RT *>(ref), shapes, loc, attn);
    else
        hipLaunchKernelGGL((prologue_fwd_kernel<T, RT, 0>), grid, dim3(256), 0, st, d, static_cast<const T *>(offsets),
                           static_cast<const T *>(logits), static_cast<const RT *>(ref), shapes, loc, attn);
    return hipGetLastError();
}

template <typename T, typename RT>
hipError_t launch_bwd(const PrologueDims &d, const void *offsets, const void *ref, const int64_t *shapes, const float *attn,
                      const float *g_loc, const float *g_attn, void *g_offsets, void *g_logits, float *g_ref, dim3 grid, hipStream_t st)
{
    const bool pitched = d.po != 0 || d.pl != 0;
    if (pitched && !(d.L == 4 && d.P == 4 && aligned16_all({offsets, attn, g_loc, g_attn, g_offsets, g_logits}) && (d.po * sizeof(T)) % 16 == 0 && (d.pl * sizeof(T)) % 16 == 0))
        return hipErrorNotSupported;
    if (d.L == 4 && d.P == 4 && aligned16_all({offsets, attn, g_loc, g_attn, g_offsets, g_logits}))
        hipLaunchKernelGGL((prologue_bwd_vec44<T, RT>), grid, dim3(256), 0, st, d, static_cast<const T *>(offsets),
                           static_cast<const RT *>(ref), shapes, attn, g_loc, g_attn, static_cast<T *>(g_offsets),
                           static_cast<T *>(g_logits), g_ref);
    else if (d.L * d.P == 16)
        hipLaunchKernelGGL((prologue_bwd_kernel<T, RT, 16>), grid, dim3(256), 0, st, d, static_cast<const T *>(offsets),
                           static_cast<const RT *>(ref), shapes, attn, g_loc, g_attn, static_cast<T *>(g_offsets),
                           static_cast<T *>(g_logits), g_ref);
    else
        hipLaunchKernelGGL((prologue_bwd_kernel<T, RT, 0>), grid, dim3(256), 0, st, d, static_cast<const T *>(offsets),
                           static_cast<const RT *>(ref), shapes, attn, g_loc, g_attn, static_cast<T *>(g_offsets),
                           static_cast<T *>(g_logits), g_ref);
    return hipGetLastError();
}

hipError_t msda_prologue_forward_launch(int io_dtype, int ref_dtype, const PrologueDims &d, const void *offsets, const void *logits,
                                        const void *ref, const int64_t *shapes, float *loc, float *attn, hipStream_t st)
{
    using BF = __hip_bfloat16;
    const int64_t total = static_cast<int64_t>(d.B) * d.Lq * d.M;
    if (total == 0) return hipSuccess;
    const dim3 grid(static_cast<unsigned>((total + 255) / 256));
    if (io_dtype == 2)
        return ref_dtype == 2 ? launch_fwd<BF, BF>(d, offsets, logits, ref, shapes, loc, attn, grid, st)
                              : launch_fwd<BF, float>(d, offsets, logits, ref, shapes, loc, attn, grid, st);
    return ref_dtype == 2 ? launch_fwd<float, BF>(d, offsets, logits, ref, shapes, loc, attn, grid, st)
                          : launch_fwd<float, float>(d, offsets, logits, ref, shapes, loc, attn, grid, st);
}

hipError_t msda_prologue_backward_launch(int io_dtype, int ref_dtype, const PrologueDims &d, const void *offsets, const void *ref,
                                         const int64_t *shapes, const float *attn, const float *g_loc, const float *g_attn,
                                         void *g_offsets, void *g_logits, float *g_ref, hipStream_t st)
{
    using BF = __hip_bfloat16;
    const int64_t total = static_cast<int64_t>(d.B) * d.Lq * d.M;
    if (total == 0) return hipSuccess;
    const dim3 grid(static_cast<unsigned>((total + 255) / 256));
    if (io_dtype == 2)
        return ref_dtype == 2 ? launch_bwd<BF, BF>(d, offsets, ref, shapes, attn, g_loc, g_attn, g_offsets, g_logits, g_ref, grid, st)
                              : launch_bwd<BF, float>(d, offsets, ref, shapes, attn, g_loc, g_attn, g_offsets, g_logits, g_ref, grid, st);
    return ref_dtype == 2 ? launch_bwd<float, BF>(d, offsets, ref, shapes, attn, g_loc, g_attn, g_offsets, g_logits, g_ref, grid, st)
                          : launch_bwd<float, float>(d, offsets, ref, shapes, attn, g_loc, g_attn, g_offsets, g_logits, g_ref, grid, st);
}

}  // namespace mdetr
