// monodetr_amd/csrc/small_wgrad.hip -- weight and bias gradient of a linear layer over a FEW thousand token rows:
//   dW[n, k] = sum_t dY[t, n] X[t, k],   db[n] = sum_t dY[t, n],   T <= 8 192, k a multiple of 64, n k <= 524 288.
// Any n: widths that are not multiples of 64 (the prediction heads' last layers: 2, 3, 6, 24 outputs; their packed first
// layers: 1 032) or dY rows that are not 16-byte aligned take guarded scalar loads of dY -- the library runs those weight
// gradients as ONE 30-47 us single-tile GEMM each.
//
// The decoder's linear layers see B x 550 = 4 400 query rows (reference depthaware_transformer.py:399-456: projections of
// both attentions, the deformable attention's offset / weight / output layers, the FFN; monodetr.py:222-262: the hidden
// layers of the prediction heads) -- 33 weight gradients of [256, 4 400] x [4 400, 256] per training step.  The library
// runs that shape as ONE 16-workgroup GEMM (34 us) or, split along the tokens, as a batched product that it handles no
// better (29 us) plus a chunk sum, and the bias gradient is another reduction over dY (2 launches): ~45 us and 4-5
// launches per layer for 0.6 GFLOP and 4.5 MB.
// Here one launch produces per-chunk partial tiles of dW AND the column sums of dY (the tiles read dY anyway); colsum.hip
// adds the chunks in a fixed order (deterministic) and rounds once into the parameter dtype.
//   grid (k / 64, n / 64, chunks); 256 threads; thread (tn, tk) owns a 4 x 4 patch of the 64 x 64 tile; 32 token rows at a
//   time are widened to fp32 into LDS ([row][64] for both operands), then per row two 16-byte LDS reads feed 8 packed
//   FMAs.  fp32 accumulation throughout (what the library's bf16 GEMM does, without its bf16 partial products).
// Not a matrix-core kernel on purpose: both MFMA operands would have to be transposed (the contraction index is the ROW of
// both matrices) and at 0.6 GFLOP the packed-FMA rate already finishes in a few microseconds.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdlib.h>

#include <mdetr_wave.h>

#include "colsum.h"
#include "small_wgrad.h"

namespace mdetr {
namespace {

constexpr int kThreads = 256;
constexpr int kTile = 64;            // output tile edge
constexpr int kRows = 32;            // token rows staged per step
constexpr int64_t kMaxRows = 8192;   // above: the library's batched split is faster (profiles/r02p_wgrad_timing.txt)
constexpr int kMaxTileArea = 2048 * 256; // n * k: the chunk partials are chunks * n * k * 4 bytes (chunks shrink as tiles grow: ~19 MB at most)
constexpr int64_t kMaxRowsNarrow = 65536;   // n <= 64: one tile row, the library has no good kernel at any height

// 8 consecutive elements of a row as loaded (native register vectors: they stay in registers across the LDS pass)
template <typename T> struct Row8;
template <> struct Row8<float> {
    struct Raw { f32x4 a, b; };
    static __device__ __forceinline__ Raw load(const float *p) { return {*reinterpret_cast<const f32x4 *>(p), *reinterpret_cast<const f32x4 *>(p + 4)}; }
    static __device__ __forceinline__ Raw zero() { f32x4 z; z.x = z.y = z.z = z.w = 0.f; return {z, z}; }
    static __device__ __forceinline__ void widen(const Raw &r, float (&v)[8])
    {
        v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
    }
};
template <> struct Row8<__hip_bfloat16> {
    typedef u32x4 Raw;
    static __device__ __forceinline__ Raw load(const __hip_bfloat16 *p) { return *reinterpret_cast<const u32x4 *>(p); }
    static __device__ __forceinline__ Raw zero() { u32x4 z; z.x = z.y = z.z = z.w = 0u; return z; }
    static __device__ __forceinline__ void widen(const Raw &u, float (&v)[8])
    {
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
        }
    }
};

__host__ __device__ inline int64_t partial_pitch(int n, int k) { return (static_cast<int64_t>(n) * k + n + 3) / 4 * 4; }

// 8 consecutive dY elements of a row in the guarded form: columns >= n read as zero, no alignment assumed
template <typename T>
__device__ __forceinline__ void load_guarded(const T *row, int col0, int n, float (&v)[8])
{
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = col0 + i < n ? static_cast<float>(row[col0 + i]) : 0.f;
}

// partial: [chunks][partial_pitch(n, k)] fp32: n * k weight-gradient entries, then n bias-gradient entries
// NGEN: any n / any dY row alignment (guarded scalar loads of dY, guarded stores)
template <typename T, bool NGEN>
__global__ __launch_bounds__(kThreads)
void small_wgrad_kernel(const T *__restrict__ dy, const T *__restrict__ x, float *__restrict__ partial, int64_t rows, int n, int k,
                        int64_t ldy, int64_t ldx, int chunk_rows)
{
    __shared__ __attribute__((aligned(16))) float sy[kRows][kTile];
    __shared__ __attribute__((aligned(16))) float sx[kRows][kTile];
    const int tile_k = blockIdx.x, tile_n = blockIdx.y, chunk = blockIdx.z;
    const int tn = threadIdx.x >> 4, tk = threadIdx.x & 15;          // 16 x 16 threads, 4 x 4 outputs each
    const int lr = threadIdx.x >> 3, lv = threadIdx.x & 7;           // staging: row 0..31, 8-element vector 0..7
    const int64_t t0 = static_cast<int64_t>(chunk) * chunk_rows;
    const int64_t t1 = t0 + chunk_rows < rows ? t0 + chunk_rows : rows;
    f32x2 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = make_f32x2(0.f, 0.f);
    float colsum[4] = {0.f, 0.f, 0.f, 0.f};
    const T *py = dy + static_cast<int64_t>(tile_n) * kTile + lv * 8;
    const T *px = x + static_cast<int64_t>(tile_k) * kTile + lv * 8;
    // the next step's rows are requested before the current step is consumed (registers carry them across the LDS pass)
    typename Row8<T>::Raw ry = Row8<T>::zero(), rx = Row8<T>::zero();
    float gy[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // NGEN: the prefetched dY elements, already widened
    const int col0 = tile_n * kTile + lv * 8;
    if (t0 + lr < t1) {
        if constexpr (NGEN) load_guarded<T>(dy + (t0 + lr) * ldy, col0, n, gy);
        else ry = Row8<T>::load(py + (t0 + lr) * ldy);
        rx = Row8<T>::load(px + (t0 + lr) * ldx);
    }
    for (int64_t base = t0; base < t1; base += kRows) {
        {
            float vy[8], vx[8];
            if constexpr (NGEN) {
#pragma unroll
                for (int i = 0; i < 8; ++i) vy[i] = gy[i];
            } else {
                Row8<T>::widen(ry, vy);
            }
            Row8<T>::widen(rx, vx);
            *reinterpret_cast<float4 *>(&sy[lr][lv * 8]) = make_float4(vy[0], vy[1], vy[2], vy[3]);
            *reinterpret_cast<float4 *>(&sy[lr][lv * 8 + 4]) = make_float4(vy[4], vy[5], vy[6], vy[7]);
            *reinterpret_cast<float4 *>(&sx[lr][lv * 8]) = make_float4(vx[0], vx[1], vx[2], vx[3]);
            *reinterpret_cast<float4 *>(&sx[lr][lv * 8 + 4]) = make_float4(vx[4], vx[5], vx[6], vx[7]);
        }
        __syncthreads();
        {
            const int64_t r = base + kRows + lr;
            const bool more = r < t1;
            if constexpr (NGEN) {
                if (more) load_guarded<T>(dy + r * ldy, col0, n, gy);
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) gy[i] = 0.f;
                }
            } else {
                ry = more ? Row8<T>::load(py + r * ldy) : Row8<T>::zero();
            }
            rx = more ? Row8<T>::load(px + r * ldx) : Row8<T>::zero();
        }
#pragma unroll 8
        for (int r = 0; r < kRows; ++r) {
            const float4 a = *reinterpret_cast<const float4 *>(&sy[r][tn * 4]);
            const float4 b = *reinterpret_cast<const float4 *>(&sx[r][tk * 4]);
            const f32x2 b01 = make_f32x2(b.x, b.y), b23 = make_f32x2(b.z, b.w);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 ai = make_f32x2(av[i], av[i]);
                acc[i][0] = fma2(ai, b01, acc[i][0]);
                acc[i][1] = fma2(ai, b23, acc[i][1]);
                colsum[i] += av[i];
            }
        }
        __syncthreads();
    }
    float *out = partial + static_cast<int64_t>(chunk) * partial_pitch(n, k);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = tile_n * kTile + tn * 4 + i;
        if (!NGEN || row < n)
            *reinterpret_cast<float4 *>(out + static_cast<int64_t>(row) * k + tile_k * kTile + tk * 4) =
                make_float4(acc[i][0].x, acc[i][0].y, acc[i][1].x, acc[i][1].y);
    }
    if (tile_k == 0 && tk == 0) {                                   // the first k tile's blocks also own the bias gradient
        float *ob = out + static_cast<int64_t>(n) * k + tile_n * kTile + tn * 4;
        if constexpr (NGEN) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tile_n * kTile + tn * 4 + i < n) ob[i] = colsum[i];
        } else {
            *reinterpret_cast<float4 *>(ob) = make_float4(colsum[0], colsum[1], colsum[2], colsum[3]);
        }
    }
    if (NGEN && chunk == 0 && tile_k == 0 && tile_n == 0 && threadIdx.x < 4) {   // the pad of the row (summed along with the rest)
        const int64_t at = static_cast<int64_t>(n) * k + n + threadIdx.x;
        if (at < partial_pitch(n, k))
            for (int c = 0; c < gridDim.z; ++c) partial[static_cast<int64_t>(c) * partial_pitch(n, k) + at] = 0.f;
    }
}

}  // namespace

bool small_wgrad_supported(int io_dtype, int64_t rows, int n, int k, int64_t ldy, int64_t ldx)
{
    return (io_dtype == 0 || io_dtype == 2) && rows > 0 && (rows <= kMaxRows || (n <= kTile && rows <= kMaxRowsNarrow)) && n > 0 && k > 0 &&
           k % kTile == 0 && static_cast<int64_t>(n) * k <= kMaxTileArea && ldy >= n && ldx >= k && ldx % 8 == 0;
}

// chunks along the token axis: ~3 workgroups per CU (a workgroup waits on one global round trip per 32 rows; several per
// CU keep the FMA pipes fed), at least 64 rows per chunk
int small_wgrad_chunks(int64_t rows, int n, int k)
{
    const int tiles = ((n + kTile - 1) / kTile) * (k / kTile);
    int target = 768;                                                        // (1152 until round 3: r03t sweep, 24.4 -> 21.3 us for the 256 x 256 layers)
    int c = (target + tiles - 1) / tiles;
    const int64_t most = (rows + 63) / 64;
    if (c > most) c = static_cast<int>(most);
    if (c > 256) c = 256;                                            // one colsum row block adds them in a single launch
    return c < 1 ? 1 : c;
}

int64_t small_wgrad_workspace_bytes(int64_t rows, int n, int k)
{
    if (!small_wgrad_supported(0, rows, n, k, n, k)) return 0;
    return static_cast<int64_t>(small_wgrad_chunks(rows, n, k)) * partial_pitch(n, k) * 4 + 64;
}

hipError_t small_wgrad_launch(int io_dtype, const void *dy, const void *x, void *out, void *workspace, int64_t rows, int n, int k,
                              int64_t ldy, int64_t ldx, int out_dtype, hipStream_t st)
{
    const int chunks = small_wgrad_chunks(rows, n, k);
    int chunk_rows = static_cast<int>((rows + chunks - 1) / chunks);
    chunk_rows = (chunk_rows + kRows - 1) / kRows * kRows;          // whole staging steps (the last chunk may be short or empty)
    float *partial = static_cast<float *>(workspace);
    const dim3 grid(static_cast<unsigned>(k / kTile), static_cast<unsigned>((n + kTile - 1) / kTile), static_cast<unsigned>(chunks));
    // the guarded form: a ragged last tile of n, or dY rows / base that 16-byte loads cannot take
    const bool ngen = n % kTile != 0 || ldy % 8 != 0 || (reinterpret_cast<uintptr_t>(dy) & 15) != 0;
    if (io_dtype == 2) {
        const auto *a = static_cast<const __hip_bfloat16 *>(dy), *b = static_cast<const __hip_bfloat16 *>(x);
        if (ngen) hipLaunchKernelGGL((small_wgrad_kernel<__hip_bfloat16, true>), grid, dim3(kThreads), 0, st, a, b, partial, rows, n, k, ldy, ldx, chunk_rows);
        else hipLaunchKernelGGL((small_wgrad_kernel<__hip_bfloat16, false>), grid, dim3(kThreads), 0, st, a, b, partial, rows, n, k, ldy, ldx, chunk_rows);
    } else {
        const auto *a = static_cast<const float *>(dy), *b = static_cast<const float *>(x);
        if (ngen) hipLaunchKernelGGL((small_wgrad_kernel<float, true>), grid, dim3(kThreads), 0, st, a, b, partial, rows, n, k, ldy, ldx, chunk_rows);
        else hipLaunchKernelGGL((small_wgrad_kernel<float, false>), grid, dim3(kThreads), 0, st, a, b, partial, rows, n, k, ldy, ldx, chunk_rows);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int cols = static_cast<int>(partial_pitch(n, k));          // `out` holds this many elements
    // <= 256 chunk rows: colsum's single-row-block form, one launch, result in out_dtype
    return colsum_launch(0, partial, out, nullptr, chunks, cols, cols, st, out_dtype);
}

}  // namespace mdetr
