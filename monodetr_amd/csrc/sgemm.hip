// monodetr_amd/csrc/sgemm.hip -- grouped fp32 products on the f32-input matrix instruction (v_mfma_f32_32x32x2_f32).
//
// The prediction heads of MonoDETR (lib/models/monodetr/monodetr.py:222-262: class / box / size / angle / depth MLPs on the
// [B x 550, 256] output of every decoder level) stay in fp32 behind the bf16 body.  They are 18 small products per level and direction
// -- [4 400, 256] x [256, 256], x [256, 6], x [256, 3] ... -- which the framework hands to the library one launch at a time (60
// launches, 0.96 ms of the round-5 step, most of it fill and drain).  This file runs them as GROUPS: one launch covers every product
// of a dependency level (the five first layers; the second layers; ...), a workgroup looks its tile up in the group's table.
//
// Arithmetic: `mfma_f32` is exact fp32 (the k-ordered fmaf chain), at the fp32 vector rate -- 1/16 of the bf16 instruction, which is
// still 155 TFLOP/s; the heads are 27 GFLOP per step.  No operand is rounded to a narrower type anywhere (bf16 OPERANDS -- the
// decoder output of the bf16 body -- are widened on load, which is exact).
//
// Layout of a product C[M, N] = sum_t A_t op(B_t):
//   NT / NN: a 256-thread workgroup owns a 64 x 64 tile, its four waves a 32 x 32 quadrant each; the contraction runs in slabs of 32
//            through LDS (double-buffered; the next slab's global loads fly during the current slab's 16 matrix instructions).  An
//            operand whose contraction index is contiguous in memory (A always; B in NT) is staged [row][32 + 4] and read with one
//            ds_read_b128 per four instructions (lane (i, h) takes k = 8 q + 4 h + 0..3: both operands use the same assignment, so the
//            products pair up correctly whatever the order); NN's B is staged [k][64 + 4] and read with ds_read_b32.
//   TN:      C[M, N] = A[K, M]^T B[K, N] contracts over the ROW index of both operands (K = 4 400 tokens): lane (i, h) loads
//            A[t + h][m0 + i] and B[t + h][n0 + i] straight from global memory -- 32 consecutive floats per row, the instruction's
//            operand layout as it lies in memory, no LDS.  A 512-thread workgroup owns a 32 x 32 tile; its eight waves take an eighth
//            of the rows each and are summed through LDS in wave order (deterministic); the column sums of A (the bias gradient) ride
//            on the A operand's registers.
// Ragged shapes are guarded per element; vector loads are used where the stride and base pointer allow them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mdetr_wave.h>

#include "msda.h"        // ProfileScope
#include "sgemm.h"

namespace mdetr {
namespace {

constexpr int kTile = 64, kSlab = 32, kPitchK = kSlab + 4, kPitchN = kTile + 4, kTnTile = 32, kTnWaves = 8;
constexpr int kMaxTerms = 24;                        // terms of all problems of a group, pooled
constexpr int kFlagCBf16 = 1, kFlagResBf16 = 2;

struct Term {
    const void *a, *b;
    int64_t lda, ldb;
    int K, a_bf16, b_bf16, pad_;
};
struct Prob {
    void *c;
    const float *bias;
    float *colsum;
    const float *mask;
    const void *res;
    int64_t ldc, ldm, ldr;
    int M, N, relu_cols, term0, nterm, tile0, tiles_n, flags;
};
struct Args {
    Term t[kMaxTerms];
    Prob p[MDETR_SGEMM_MAX_PROBLEMS];
    int nprob;
};

__device__ __forceinline__ float bf16_bits_to_float(unsigned short v) { return __uint_as_float(static_cast<unsigned>(v) << 16); }

// four consecutive elements starting at element offset `off`; only the first `valid` (<= 4) exist, the rest read as zero
__device__ __forceinline__ f32x4 load4(const void *base, int bf16, int64_t off, int valid, bool vec)
{
    f32x4 r;
    r.x = r.y = r.z = r.w = 0.f;
    if (valid <= 0) return r;
    if (bf16) {
        const unsigned short *p = static_cast<const unsigned short *>(base) + off;
        if (vec && valid >= 4) {
            const uint2 v = *reinterpret_cast<const uint2 *>(p);
            r.x = __uint_as_float(v.x << 16); r.y = __uint_as_float(v.x & 0xffff0000u);
            r.z = __uint_as_float(v.y << 16); r.w = __uint_as_float(v.y & 0xffff0000u);
        } else {
            r.x = bf16_bits_to_float(p[0]);
            if (valid > 1) r.y = bf16_bits_to_float(p[1]);
            if (valid > 2) r.z = bf16_bits_to_float(p[2]);
            if (valid > 3) r.w = bf16_bits_to_float(p[3]);
        }
    } else {
        const float *p = static_cast<const float *>(base) + off;
        if (vec && valid >= 4) {
            r = *reinterpret_cast<const f32x4 *>(p);
        } else {
            r.x = p[0];
            if (valid > 1) r.y = p[1];
            if (valid > 2) r.z = p[2];
            if (valid > 3) r.w = p[3];
        }
    }
    return r;
}

__device__ __forceinline__ float load1(const void *base, int bf16, int64_t off)
{
    return bf16 ? bf16_bits_to_float(static_cast<const unsigned short *>(base)[off]) : static_cast<const float *>(base)[off];
}

// can rows of this operand be read with one 8- / 16-byte load per four elements?  (base and row stride aligned; the column offsets the
// kernels use are multiples of four)
__device__ __forceinline__ bool vec_ok(const void *base, int64_t ld, int bf16)
{
    const uintptr_t mask = bf16 ? 7u : 15u;
    return (reinterpret_cast<uintptr_t>(base) & mask) == 0 && (ld & 3) == 0;
}

__device__ __forceinline__ int clamp04(int v) { return v < 0 ? 0 : (v > 4 ? 4 : v); }

__device__ __forceinline__ int find_problem(const Args &a, int tile)
{
    int pi = 0;
    for (int i = 1; i < a.nprob; ++i)
        if (tile >= a.p[i].tile0) pi = i;
    return pi;
}

__device__ __forceinline__ void epilogue_store(const Prob &P, int gm, int gn, float v)
{
    if (gm >= P.M || gn >= P.N) return;
    if (P.bias) v += P.bias[gn];
    if (P.res) v += load1(P.res, P.flags & kFlagResBf16, static_cast<int64_t>(gm) * P.ldr + gn);
    if (gn < P.relu_cols) v = fmaxf(v, 0.f);
    if (P.mask && !(P.mask[static_cast<int64_t>(gm) * P.ldm + gn] > 0.f)) v = 0.f;
    const int64_t o = static_cast<int64_t>(gm) * P.ldc + gn;
    if (P.flags & kFlagCBf16) static_cast<__bf16 *>(P.c)[o] = static_cast<__bf16>(v);
    else static_cast<float *>(P.c)[o] = v;
}

// ---- NT / NN -----------------------------------------------------------------------------------------------------------------
template <bool NN>
__global__ __launch_bounds__(256)
void sgemm_kernel(const Args a)
{
    __shared__ __attribute__((aligned(16))) float As[2][kTile * kPitchK];
    __shared__ __attribute__((aligned(16))) float Bs[2][NN ? kSlab * kPitchN : kTile * kPitchK];
    const int tile = blockIdx.x;
    const Prob &P = a.p[find_problem(a, tile)];
    const int lt = tile - P.tile0, tm = lt / P.tiles_n, tn = lt - tm * P.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;

    // staging maps: K-contiguous operand: row = tid / 8 (+ 32), 4 columns from (tid % 8) * 4; NN's B: k = tid / 16 (+ 16), 4 columns from (tid % 16) * 4
    const int sr = tid >> 3, sc = (tid & 7) << 2;
    const int br = NN ? tid >> 4 : sr, bc = NN ? (tid & 15) << 2 : sc;

    int total = 0;
    for (int t = 0; t < P.nterm; ++t) total += (a.t[P.term0 + t].K + kSlab - 1) / kSlab;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    f32x4 ra[2], rb[2];
    int ti = 0, k0 = 0;                                            // the slab to load next: term, offset
    auto load_slab = [&]() {
        const Term &T = a.t[P.term0 + ti];
        const bool va = vec_ok(T.a, T.lda, T.a_bf16), vb = vec_ok(T.b, T.ldb, T.b_bf16);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = tm * kTile + sr + 32 * p, k = k0 + sc;
            ra[p] = load4(T.a, T.a_bf16, static_cast<int64_t>(row) * T.lda + k, row < P.M ? clamp04(T.K - k) : 0, va);
            if (NN) {
                const int kk = k0 + br + 16 * p, col = tn * kTile + bc;
                rb[p] = load4(T.b, T.b_bf16, static_cast<int64_t>(kk) * T.ldb + col, kk < T.K ? clamp04(P.N - col) : 0, vb);
            } else {
                const int rowb = tn * kTile + br + 32 * p;
                rb[p] = load4(T.b, T.b_bf16, static_cast<int64_t>(rowb) * T.ldb + k, rowb < P.N ? clamp04(T.K - k) : 0, vb);
            }
        }
        k0 += kSlab;
        if (k0 >= T.K) { k0 = 0; ++ti; }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<f32x4 *>(&As[buf][(sr + 32 * p) * kPitchK + sc]) = ra[p];
            if (NN) *reinterpret_cast<f32x4 *>(&Bs[buf][(br + 16 * p) * kPitchN + bc]) = rb[p];
            else *reinterpret_cast<f32x4 *>(&Bs[buf][(br + 32 * p) * kPitchK + bc]) = rb[p];
        }
    };

    if (total > 0) {
        load_slab();
        store_slab(0);
    }
    __syncthreads();
    for (int s = 0; s < total; ++s) {
        const int buf = s & 1;
        if (s + 1 < total) load_slab();
        const float *ap = &As[buf][(wm * 32 + li) * kPitchK + 4 * lh];
        if (NN) {
            const float *bp = &Bs[buf][(4 * lh) * kPitchN + wn * 32 + li];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 8 * q);
                acc = mfma_f32(av.x, bp[(8 * q + 0) * kPitchN], acc);
                acc = mfma_f32(av.y, bp[(8 * q + 1) * kPitchN], acc);
                acc = mfma_f32(av.z, bp[(8 * q + 2) * kPitchN], acc);
                acc = mfma_f32(av.w, bp[(8 * q + 3) * kPitchN], acc);
            }
        } else {
            const float *bp = &Bs[buf][(wn * 32 + li) * kPitchK + 4 * lh];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 8 * q);
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(bp + 8 * q);
                acc = mfma_f32(av.x, bv.x, acc);
                acc = mfma_f32(av.y, bv.y, acc);
                acc = mfma_f32(av.z, bv.z, acc);
                acc = mfma_f32(av.w, bv.w, acc);
            }
        }
        if (s + 1 < total) store_slab(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        epilogue_store(P, tm * kTile + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, tn * kTile + wn * 32 + li, acc[r]);
}

// ---- TN ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kTnWaves, 4)
void sgemm_tn_kernel(const Args a)
{
    __shared__ float red[kTnWaves - 1][16 * 64];
    __shared__ float cred[kTnWaves][32];
    const int tile = blockIdx.x;
    const Prob &P = a.p[find_problem(a, tile)];
    const Term &T = a.t[P.term0];
    const int lt = tile - P.tile0, tm = lt / P.tiles_n, tn = lt - tm * P.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int m = tm * kTnTile + li, n = tn * kTnTile + li;
    const bool okm = m < P.M, okn = n < P.N;
    // this wave's rows of the contraction: an even count per wave so that the two lane halves pair up inside it
    const int chunk = ((T.K + 2 * kTnWaves - 1) / (2 * kTnWaves)) * 2;
    const int t_begin = wave * chunk, t_end = min(T.K, t_begin + chunk);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float csum = 0.f;

    constexpr int E = 16;                                          // matrix instructions per step: 2 E rows of the contraction
    float av[E], bv[E], an[E], bn[E];
    auto load_step = [&](int t0, float (&x)[E], float (&y)[E]) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int t = t0 + 2 * e + lh;
            const bool ok = t < t_end;
            x[e] = (ok && okm) ? load1(T.a, T.a_bf16, static_cast<int64_t>(t) * T.lda + m) : 0.f;
            y[e] = (ok && okn) ? load1(T.b, T.b_bf16, static_cast<int64_t>(t) * T.ldb + n) : 0.f;
        }
    };
    if (t_begin < t_end) load_step(t_begin, av, bv);
    for (int t0 = t_begin; t0 < t_end; t0 += 2 * E) {
        const bool more = t0 + 2 * E < t_end;
        if (more) load_step(t0 + 2 * E, an, bn);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            acc = mfma_f32(av[e], bv[e], acc);
            csum += av[e];
        }
        if (more) {
#pragma unroll
            for (int e = 0; e < E; ++e) { av[e] = an[e]; bv[e] = bn[e]; }
        }
    }
    // the eight partial tiles, summed in wave order
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r * 64 + lane] = acc[r];
    }
    csum += __shfl_xor(csum, 32);
    if (lh == 0) cred[wave][li] = csum;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 1
    for (int w = 0; w < kTnWaves - 1; ++w) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[w][r * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        epilogue_store(P, tm * kTnTile + (r & 3) + 8 * (r >> 2) + 4 * lh, tn * kTnTile + li, acc[r]);
    if (P.colsum && tn == 0 && lh == 0 && okm) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < kTnWaves; ++w) s += cred[w][li];
        P.colsum[m] = s;
    }
}

bool dtype_ok(int d) { return d == MDETR_F32 || d == MDETR_BF16; }

}  // namespace

const char *sgemm_check(int mode, const mdetr_sgemm_problem *p, int nprob)
{
    if (mode != MDETR_SGEMM_NT && mode != MDETR_SGEMM_NN && mode != MDETR_SGEMM_TN) return "unknown mode";
    if (!p || nprob <= 0 || nprob > MDETR_SGEMM_MAX_PROBLEMS) return "1 .. MDETR_SGEMM_MAX_PROBLEMS problems per group";
    int terms = 0;
    for (int i = 0; i < nprob; ++i) {
        const mdetr_sgemm_problem &q = p[i];
        if (q.m <= 0 || q.n <= 0 || q.nterm <= 0 || q.nterm > MDETR_SGEMM_MAX_TERMS) return "bad shape (m, n > 0; 1 .. MDETR_SGEMM_MAX_TERMS terms)";
        if (!q.c || q.ldc < q.n || !dtype_ok(q.c_dtype)) return "bad result (null, row stride < n, or element type)";
        if (q.relu_cols < 0 || q.relu_cols > q.n) return "relu_cols outside [0, n]";
        if (q.mask && q.ldm < q.n) return "mask row stride < n";
        if (q.res && (q.ldr < q.n || !dtype_ok(q.res_dtype))) return "bad res (row stride < n, or element type)";
        if (mode == MDETR_SGEMM_TN && (q.nterm != 1 || q.bias || q.mask || q.res || q.relu_cols)) return "TN: one term, no bias / relu / mask / res";
        if (mode != MDETR_SGEMM_TN && q.colsum) return "colsum belongs to TN";
        for (int t = 0; t < q.nterm; ++t) {
            const mdetr_sgemm_term &x = q.term[t];
            if (!x.a || !x.b || x.k <= 0 || !dtype_ok(x.a_dtype) || !dtype_ok(x.b_dtype)) return "bad term (null operand, k <= 0, or element type)";
            const int64_t a_cols = mode == MDETR_SGEMM_TN ? q.m : x.k, b_cols = mode == MDETR_SGEMM_NT ? x.k : q.n;
            if (x.lda < a_cols || x.ldb < b_cols) return "operand row stride shorter than its row";
        }
        terms += q.nterm;
    }
    if (terms > kMaxTerms) return "too many terms in one group";
    return nullptr;
}

hipError_t sgemm_launch(int mode, const mdetr_sgemm_problem *p, int nprob, hipStream_t st)
{
    Args a;
    int tiles = 0, terms = 0;
    double flop = 0.0, bytes = 0.0;
    const int tsz = mode == MDETR_SGEMM_TN ? kTnTile : kTile;
    for (int i = 0; i < nprob; ++i) {
        const mdetr_sgemm_problem &q = p[i];
        Prob &P = a.p[i];
        P.c = q.c; P.bias = q.bias; P.colsum = q.colsum; P.mask = q.mask; P.res = q.res;
        P.ldc = q.ldc; P.ldm = q.ldm; P.ldr = q.ldr;
        P.M = q.m; P.N = q.n; P.relu_cols = q.relu_cols; P.term0 = terms; P.nterm = q.nterm;
        P.tile0 = tiles; P.tiles_n = (q.n + tsz - 1) / tsz;
        P.flags = (q.c_dtype == MDETR_BF16 ? kFlagCBf16 : 0) | (q.res && q.res_dtype == MDETR_BF16 ? kFlagResBf16 : 0);
        tiles += ((q.m + tsz - 1) / tsz) * P.tiles_n;
        bytes += static_cast<double>(q.m) * q.n * (q.c_dtype == MDETR_BF16 ? 2.0 : 4.0) * (1.0 + (q.res ? 1.0 : 0.0)) + (q.mask ? 4.0 * q.m * q.n : 0.0);
        for (int t = 0; t < q.nterm; ++t) {
            const mdetr_sgemm_term &x = q.term[t];
            Term &T = a.t[terms++];
            T.a = x.a; T.b = x.b; T.lda = x.lda; T.ldb = x.ldb; T.K = x.k;
            T.a_bf16 = x.a_dtype == MDETR_BF16; T.b_bf16 = x.b_dtype == MDETR_BF16; T.pad_ = 0;
            flop += 2.0 * q.m * q.n * x.k;
            bytes += static_cast<double>(x.k) * (q.m * (T.a_bf16 ? 2.0 : 4.0) + q.n * (T.b_bf16 ? 2.0 : 4.0));
        }
    }
    a.nprob = nprob;
    ProfileScope prof(21, tiles, st, flop / 1e6, bytes / 1e3);
    if (mode == MDETR_SGEMM_TN) hipLaunchKernelGGL(sgemm_tn_kernel, dim3(tiles), dim3(64 * kTnWaves), 0, st, a);
    else if (mode == MDETR_SGEMM_NN) hipLaunchKernelGGL(sgemm_kernel<true>, dim3(tiles), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(sgemm_kernel<false>, dim3(tiles), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace mdetr
