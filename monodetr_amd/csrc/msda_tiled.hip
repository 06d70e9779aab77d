// monodetr_amd/csrc/msda_tiled.hip -- grad_value of MSDA for the self-attention (encoder) case,
// without one global atomic per contribution.
//
// Why: the backward scatter `grad_value[b, pix, m, :] += w * attn * grad_out[b, q, m, :]`
// (reference: atomicAdd per corner and channel, ms_deform_im2col_cuda.cuh:125-152) issues
// B*Lq*M*L*P*4*D = 1.34 G fp32 atomics for the encoder call.  Measured on MI355X
// (tools/ubench/atomics.hip): the L2 atomic unit retires ~320 G fp32 atomics/s whatever the
// address pattern (=> >= 4 ms), LDS `ds_add_f32` is even slower (0.2 T/s), but LDS *integer*
// atomics run at 4.5-6.9 T/s.  So:
//
//   1. absmax pre-pass: M = max|grad_out| * max|attn| (two device scalars) fixes ONE power-of-two
//      scale 2^s with |w*attn*g| * 2^s < 2^46.
//   2. scatter: a workgroup owns (image b, head m, destination level l, query tile t).  Its LDS
//      holds an int64 window [WH_l x WW_l cells][32 channels] of level l: the tile's footprint
//      plus a margin R (levels whose whole map fits are held completely; their queries are split
//      into chunks).  Every contribution is converted to 64-bit fixed point (exact f64 FMA with
//      the 1.5*2^52 trick) and added with `ds_add_u64`; integer addition is associative, so the
//      privatised sum is exact and order-independent (the fp32-atomic path is not).  Corners that
//      fall outside the window (large learned offsets) go to the reference-style global fp32
//      atomic.  Footprints are computed one SAMPLE per lane (64 per wave step) and handed to the
//      32-channel half-waves through a wave-private LDS record, instead of redundantly in all 32
//      channel lanes.
//   3. windows are written to a scratch buffer (plain coalesced stores); a reduce kernel adds, for
//      every grad_value row, the windows that cover it (geometry is static) -- no atomics.
//
// Queries are tiled by their own pyramid position (query q of the flattened pyramid sits at a
// pixel centre), which is where their samples land when offsets are local.  Arbitrary sampling
// locations stay CORRECT (fallback atomics), only slower.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <mdetr_wave.h>

#include "msda.h"

namespace mdetr {
namespace {

constexpr int kMaxLevels = 4;
constexpr int kCH = 32;              // channels per head (fast path geometry)
constexpr int kThreads = 1024;       // 16 waves per workgroup (the scatter is latency-bound: loc/attn gather -> grad_out rows -> LDS atomics)
constexpr int kWavesT = kThreads / 64;
constexpr int kRecDwords = 8;        // per-sample record, 32 B: {q:u16, pad, off[4]:i16} {w[4]:f32}
constexpr int kFixedBits = 44;       // |contribution| * scale < 2^44; < 2^19 contributions per cell fit int64

struct TilePlan {
    int B, S, M, L, P, Lq;
    int H[kMaxLevels], W[kMaxLevels], start[kMaxLevels];
    int TH[kMaxLevels], TW[kMaxLevels];      // tile core in level-l cells; 0 => whole level, query chunks
    int nty[kMaxLevels], ntx[kMaxLevels];    // tiles per axis (chunks: nty = 1, ntx = #chunks)
    int R[kMaxLevels];
    int WH[kMaxLevels], WW[kMaxLevels];      // window size in cells
    int blk0[kMaxLevels + 1];                // first block (within one (b, m)) of each level
    long long scr0[kMaxLevels];              // float offset of level l's windows within one (b, m) slab
    long long scr_per_bm;
    int max_cells;
};

// ---- tiny helpers ---------------------------------------------------------------------------------
__host__ __device__ inline int ceil_div_i(long long a, long long b)   // b > 0, any sign of a
{
    return static_cast<int>(a >= 0 ? (a + b - 1) / b : -((-a) / b));
}

// first query row y of a level with H_q rows whose centre maps to a level-l cell >= t
// (cell = floor((2y+1) * H_l / (2 H_q)))
__host__ __device__ inline int first_row_at_or_after(int t, int H_l, int H_q)
{
    const int y = ceil_div_i(2LL * t * H_q - H_l, 2LL * H_l);
    return y < 0 ? 0 : (y > H_q ? H_q : y);
}

__device__ __forceinline__ float pix_coord_f(float loc, int size)
{
#pragma clang fp contract(off)
    const float prod = loc * static_cast<float>(size);
    return prod - 0.5f;
}

// x * 2^s as 64-bit fixed point, round-to-nearest: one exact f64 FMA puts the integer into the
// low mantissa bits of 1.5*2^52 + x*2^s (valid for |x*2^s| < 2^51)
__device__ __forceinline__ unsigned long long to_fixed(float x, double scale)
{
    const double d = __builtin_fma(static_cast<double>(x), scale, 6755399441055744.0);
    const unsigned long long bits = __builtin_bit_cast(unsigned long long, d);
    const unsigned lo = static_cast<unsigned>(bits);
    const int hi = static_cast<int>((static_cast<unsigned>(bits >> 32) & 0xFFFFFu)) - 0x80000;
    return (static_cast<unsigned long long>(static_cast<unsigned>(hi)) << 32) | lo;
}

// ---- 1. absmax -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void absmax2_kernel(const float *__restrict__ a, int64_t na, const float *__restrict__ b, int64_t nb,
                    unsigned *__restrict__ out2)
{
    // na, nb are multiples of 4 (32 channels per head, L*P % 4 == 0 on the fast path); bases 16-B aligned
    float ma = 0.f, mb = 0.f;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    const float4 *a4 = reinterpret_cast<const float4 *>(a), *b4 = reinterpret_cast<const float4 *>(b);
    // max of |x| over finite and +inf values; NaN is caught through the sum below (NaN or inf-inf poison it)
    float poison = 0.f;
    for (int64_t i = tid; i < na / 4; i += stride) {
        const float4 v = a4[i];
        ma = fmaxf(fmaxf(ma, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        poison += (v.x + v.y + v.z + v.w) * 0.f;
    }
    for (int64_t i = tid; i < nb / 4; i += stride) {
        const float4 v = b4[i];
        mb = fmaxf(fmaxf(mb, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        poison += (v.x + v.y + v.z + v.w) * 0.f;
    }
    if (!(poison == 0.f)) ma = __builtin_inff();        // a NaN (or inf) somewhere -> everything takes the atomic path
    for (int o = 32; o > 0; o >>= 1) { ma = fmaxf(ma, __shfl_xor(ma, o)); mb = fmaxf(mb, __shfl_xor(mb, o)); }
    if ((threadIdx.x & 63) == 0) {                       // non-negative floats order like their bit patterns
        atomicMax(out2, __builtin_bit_cast(unsigned, ma));
        atomicMax(out2 + 1, __builtin_bit_cast(unsigned, mb));
    }
}

// ---- 2. scatter ----------------------------------------------------------------------------------
struct BlockWork { int l, kind; int y0[kMaxLevels], y1[kMaxLevels], x0[kMaxLevels], x1[kMaxLevels]; int q0, q1; int wy0, wx0; int nq; };

__device__ __forceinline__ BlockWork decode_block(const TilePlan &pl, int k)
{
    BlockWork w;
    int l = 0;
    while (l + 1 < pl.L && k >= pl.blk0[l + 1]) ++l;
    w.l = l;
    const int t = k - pl.blk0[l];
    if (pl.TH[l] == 0) {                                 // whole-level window, contiguous query chunk
        w.kind = 1;
        const int nch = pl.ntx[l];
        w.q0 = static_cast<int>(static_cast<long long>(pl.Lq) * t / nch);
        w.q1 = static_cast<int>(static_cast<long long>(pl.Lq) * (t + 1) / nch);
        w.nq = w.q1 - w.q0;
        w.wy0 = 0; w.wx0 = 0;
    } else {                                             // spatial tile: one rectangle per query level
        w.kind = 0;
        const int ty = t / pl.ntx[l], tx = t % pl.ntx[l];
        const int cy0 = ty * pl.TH[l], cy1 = cy0 + pl.TH[l], cx0 = tx * pl.TW[l], cx1 = cx0 + pl.TW[l];
        w.wy0 = cy0 - pl.R[l]; w.wx0 = cx0 - pl.R[l];
        w.nq = 0;
#pragma unroll
        for (int lq = 0; lq < kMaxLevels; ++lq) {
            if (lq >= pl.L) { w.y0[lq] = w.y1[lq] = w.x0[lq] = w.x1[lq] = 0; continue; }
            w.y0[lq] = first_row_at_or_after(cy0, pl.H[l], pl.H[lq]);
            w.y1[lq] = first_row_at_or_after(cy1, pl.H[l], pl.H[lq]);
            w.x0[lq] = first_row_at_or_after(cx0, pl.W[l], pl.W[lq]);
            w.x1[lq] = first_row_at_or_after(cx1, pl.W[l], pl.W[lq]);
            w.nq += (w.y1[lq] - w.y0[lq]) * (w.x1[lq] - w.x0[lq]);
        }
    }
    return w;
}

// i-th query of the block's set -> flattened query index
__device__ __forceinline__ int nth_query(const TilePlan &pl, const BlockWork &w, int i)
{
    if (w.kind == 1) return w.q0 + i;
    int q = 0;
    bool done = false;
#pragma unroll
    for (int lq = 0; lq < kMaxLevels; ++lq) {
        const int wx = w.x1[lq] - w.x0[lq], n = (w.y1[lq] - w.y0[lq]) * wx;
        if (!done && i < n) { q = pl.start[lq] + (w.y0[lq] + i / wx) * pl.W[lq] + w.x0[lq] + i % wx; done = true; }
        i -= n;
    }
    return q;
}

// one sample (record) x this lane's channel: up to four corner contributions
__device__ __forceinline__ void accumulate_sample(const unsigned *rr, float g, unsigned long long *win, float *gv_level,
                                                  int pixel_stride, int c, double scale)
{
    const uint4 r0 = *reinterpret_cast<const uint4 *>(rr);
    const uint4 r1 = *reinterpret_cast<const uint4 *>(rr + 4);
    const int off[4] = {static_cast<short>(r0.y & 0xFFFFu), static_cast<short>(r0.y >> 16),
                        static_cast<short>(r0.z & 0xFFFFu), static_cast<short>(r0.z >> 16)};
    const float wt[4] = {__builtin_bit_cast(float, r1.x), __builtin_bit_cast(float, r1.y),
                         __builtin_bit_cast(float, r1.z), __builtin_bit_cast(float, r1.w)};
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        if (off[cc] >= 0) {
            __hip_atomic_fetch_add(win + off[cc] * kCH + c, to_fixed(wt[cc] * g, scale), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (off[cc] <= -2) {
            unsafeAtomicAdd(gv_level + static_cast<int64_t>(-2 - off[cc]) * pixel_stride + c, wt[cc] * g);
        }
    }
}

template <typename GT>
__global__ __launch_bounds__(kThreads)
void msda_scatter_tiles(const TilePlan pl, const float *__restrict__ loc, const float *__restrict__ attn,
                        const GT *__restrict__ grad_out, float *__restrict__ grad_value,
                        const unsigned *__restrict__ absmax2, float *__restrict__ scratch)
{
    MDETR_DYNAMIC_LDS(unsigned char, smem_raw);
    unsigned long long *win = reinterpret_cast<unsigned long long *>(smem_raw);
    unsigned *recs = reinterpret_cast<unsigned *>(smem_raw + static_cast<size_t>(pl.max_cells) * kCH * 8);

    const int bid = blockIdx.x;
    const int b = bid % pl.B, r = bid / pl.B, m = r % pl.M, k = r / pl.M;
    const BlockWork w = decode_block(pl, k);
    const int l = w.l, H = pl.H[l], W = pl.W[l], WH = pl.WH[l], WW = pl.WW[l];
    const int ncell = WH * WW;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int P = pl.P, LP = pl.L * P;

    for (int i = threadIdx.x; i < ncell * kCH; i += kThreads) win[i] = 0ull;

    // one power-of-two scale per call: |w * attn * g| <= max|g| * max|attn| = mx
    const float mx = __builtin_bit_cast(float, absmax2[0]) * __builtin_bit_cast(float, absmax2[1]);
    const bool finite = mx <= 3.0e38f;                   // false for inf / NaN -> everything takes the atomic path
    int e = 0;
    if (finite && mx > 0.f) (void)frexpf(mx, &e);        // mx < 2^e
    const double scale = __builtin_ldexp(1.0, kFixedBits - e);
    __syncthreads();

    unsigned *myrec = recs + wave * 64 * kRecDwords;
    const int64_t pair_base = (static_cast<int64_t>(b) * pl.Lq) * pl.M + m;     // + q * M
    float *gv_level = grad_value + (static_cast<int64_t>(b) * pl.S + pl.start[l]) * (pl.M * kCH) + m * kCH;
    const int nsamp = w.nq * P;

    for (int base = wave * 64; base < nsamp; base += kWavesT * 64) {
        // ---- a. one sample per lane: footprint -> record -------------------------------------------
        {
            const int i = base + lane;
            int q = 0, off[4] = {-1, -1, -1, -1};
            float wt[4] = {0.f, 0.f, 0.f, 0.f};
            if (i < nsamp) {
                q = nth_query(pl, w, i / P);
                const int p = i % P;
                const int64_t rec = (pair_base + static_cast<int64_t>(q) * pl.M) * LP + l * P + p;
                const float2 xy = *reinterpret_cast<const float2 *>(loc + rec * 2);
                const float a = attn[rec];
                const float h_im = pix_coord_f(xy.y, H), w_im = pix_coord_f(xy.x, W);
                if (h_im > -1.f && w_im > -1.f && h_im < static_cast<float>(H) && w_im < static_cast<float>(W)) {   // .cuh:288
                    const float hf = floorf(h_im), wf = floorf(w_im);
                    const int y = static_cast<int>(hf), x = static_cast<int>(wf);
                    const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
                    const float cw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int yy = y + (c >> 1), xx = x + (c & 1);
                        if (yy >= 0 && yy <= H - 1 && xx >= 0 && xx <= W - 1) {          // .cuh:56-74
                            wt[c] = cw[c] * a;
                            const int wy = yy - w.wy0, wx = xx - w.wx0;
                            if (finite && wy >= 0 && wy < WH && wx >= 0 && wx < WW) off[c] = (wy * WW + wx);          // window cell
                            else off[c] = -2 - (yy * W + xx);                                                          // far: global pixel
                        }
                    }
                }
            }
            unsigned *rr = myrec + lane * kRecDwords;
            const unsigned o01 = (static_cast<unsigned>(off[0]) & 0xFFFFu) | (static_cast<unsigned>(off[1]) << 16);
            const unsigned o23 = (static_cast<unsigned>(off[2]) & 0xFFFFu) | (static_cast<unsigned>(off[3]) << 16);
            *reinterpret_cast<uint4 *>(rr) = make_uint4(static_cast<unsigned>(q), o01, o23, 0u);
            *reinterpret_cast<uint4 *>(rr + 4) = make_uint4(__builtin_bit_cast(unsigned, wt[0]), __builtin_bit_cast(unsigned, wt[1]),
                                                            __builtin_bit_cast(unsigned, wt[2]), __builtin_bit_cast(unsigned, wt[3]));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- b. 32 channel lanes per sample: fixed-point accumulate -------------------------------
        // half-wave h walks samples [32h, 32h+32) of the step; consecutive samples share their query in
        // runs of P, so grad_out rows are fetched once per query and all of a step's rows are requested
        // up front (the per-sample dependent load was the bottleneck: 2 ms -> latency-bound).
        const int c = lane & 31, half = lane >> 5;
        const int nloc = min(64, nsamp - base);
        const unsigned *hrec = myrec + half * 32 * kRecDwords;
        if (P == 4) {
            float g[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int q = static_cast<int>(hrec[(t * 4) * kRecDwords]);
                g[t] = Elem<GT>::load1(grad_out + (pair_base + static_cast<int64_t>(q) * pl.M) * kCH + c);   // q = 0 for padding records
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const int j = t * 4 + pp;
                    if (half * 32 + j < nloc) accumulate_sample(hrec + j * kRecDwords, g[t], win, gv_level, pl.M * kCH, c, scale);
                }
            }
        } else {
            for (int j = 0; j < 32; ++j) {
                if (half * 32 + j < nloc) {
                    const int q = static_cast<int>(hrec[j * kRecDwords]);
                    const float gq = Elem<GT>::load1(grad_out + (pair_base + static_cast<int64_t>(q) * pl.M) * kCH + c);
                    accumulate_sample(hrec + j * kRecDwords, gq, win, gv_level, pl.M * kCH, c, scale);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();

    // ---- c. window -> scratch (plain coalesced stores) ---------------------------------------------
    float *dst = scratch + (static_cast<int64_t>(b) * pl.M + m) * pl.scr_per_bm + pl.scr0[l] +
                 static_cast<int64_t>(k - pl.blk0[l]) * ncell * kCH;
    const double inv = 1.0 / scale;
    for (int i = threadIdx.x; i < ncell * kCH; i += kThreads)
        dst[i] = static_cast<float>(static_cast<double>(static_cast<long long>(win[i])) * inv);
}

// ---- 3. reduce: grad_value row += sum of the windows that cover it ----------------------------------
__global__ __launch_bounds__(256)
void msda_reduce_tiles(const TilePlan pl, const float *__restrict__ scratch, float *__restrict__ grad_value)
{
    // 8 lanes x float4 per (b, pixel, m) row of 32 channels
    const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    const int64_t row = gid >> 3;
    const int c4 = static_cast<int>(gid & 7) * 4;
    const int64_t nrows = static_cast<int64_t>(pl.B) * pl.S * pl.M;
    if (row >= nrows) return;
    const int m = static_cast<int>(row % pl.M);
    const int64_t bp = row / pl.M;
    const int pix = static_cast<int>(bp % pl.S), b = static_cast<int>(bp / pl.S);
    int l = 0;
    while (l + 1 < pl.L && pix >= pl.start[l + 1]) ++l;
    const int y = (pix - pl.start[l]) / pl.W[l], x = (pix - pl.start[l]) % pl.W[l];
    const int WH = pl.WH[l], WW = pl.WW[l], ncell = WH * WW;
    const float *base = scratch + (static_cast<int64_t>(b) * pl.M + m) * pl.scr_per_bm + pl.scr0[l] + c4;
    float4 acc = *reinterpret_cast<const float4 *>(grad_value + row * kCH + c4);     // far-sample atomics landed here
    auto add = [&](int64_t cell) {
        const float4 v = *reinterpret_cast<const float4 *>(base + cell * kCH);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    };
    if (pl.TH[l] == 0) {
        for (int t = 0; t < pl.ntx[l]; ++t) add(static_cast<int64_t>(t) * ncell + y * WW + x);
    } else {
        const int TH = pl.TH[l], TW = pl.TW[l], R = pl.R[l];
        const int ty0 = max(0, ceil_div_i(y - R - TH + 1, TH)), ty1 = min(pl.nty[l] - 1, (y + R) / TH);
        const int tx0 = max(0, ceil_div_i(x - R - TW + 1, TW)), tx1 = min(pl.ntx[l] - 1, (x + R) / TW);
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx)
                add(static_cast<int64_t>(ty * pl.ntx[l] + tx) * ncell + (y - (ty * TH - R)) * WW + (x - (tx * TW - R)));
    }
    *reinterpret_cast<float4 *>(grad_value + row * kCH + c4) = acc;                  // exclusive owner: plain store
}

}  // namespace

// ---- host ---------------------------------------------------------------------------------------------
namespace {

// LDS budget: windows <= 600 cells * 32 ch * 8 B = 150 KiB, + 8 waves * 64 records * 48 B = 24 KiB... too much:
// keep windows <= 512 cells (128 KiB) so that records (24 KiB) fit in the 160 KiB of a CU.
constexpr int kMaxCells = 512;

bool build_plan(TilePlan &pl, const int64_t *shapes_h, const int64_t *start_h, int B, int S, int M, int L, int Lq, int P)
{
    if (L < 1 || L > kMaxLevels || P < 1 || P > 8) return false;
    memset(&pl, 0, sizeof(pl));
    pl.B = B; pl.S = S; pl.M = M; pl.L = L; pl.P = P; pl.Lq = Lq;
    int64_t total = 0;
    for (int l = 0; l < L; ++l) {
        pl.H[l] = static_cast<int>(shapes_h[2 * l]);
        pl.W[l] = static_cast<int>(shapes_h[2 * l + 1]);
        pl.start[l] = static_cast<int>(start_h[l]);
        if (pl.H[l] <= 0 || pl.W[l] <= 0 || pl.start[l] != total) return false;
        if (static_cast<int64_t>(pl.H[l]) * pl.W[l] > 32766) return false;      // far-pixel index is packed in 16 bits
        total += static_cast<int64_t>(pl.H[l]) * pl.W[l];
    }
    if (total != S || Lq != S) return false;              // self-attention over the pyramid only
    int blk = 0;
    long long scr = 0;
    pl.max_cells = 0;
    for (int l = 0; l < L; ++l) {
        const int H = pl.H[l], W = pl.W[l];
        pl.blk0[l] = blk;
        pl.scr0[l] = scr;
        if (H * W <= kMaxCells) {                          // whole level in LDS; split the queries instead
            pl.TH[l] = pl.TW[l] = 0; pl.R[l] = 0;
            pl.WH[l] = H; pl.WW[l] = W;
            pl.nty[l] = 1;
            pl.ntx[l] = 8;                                 // query chunks
        } else {
            const int R = 4;
            int TH = 8, TW = 24;                           // (8+8) x (24+8) = 512 cells
            if (H + 0 <= 16) { TH = H; TW = kMaxCells / (H + 2 * R) - 2 * R; }
            if (TW < 4) return false;
            pl.TH[l] = TH; pl.TW[l] = TW; pl.R[l] = R;
            pl.WH[l] = TH + 2 * R; pl.WW[l] = TW + 2 * R;
            pl.nty[l] = (H + TH - 1) / TH;
            pl.ntx[l] = (W + TW - 1) / TW;
        }
        const int nb = pl.nty[l] * pl.ntx[l];
        const int cells = pl.WH[l] * pl.WW[l];
        if (cells > kMaxCells) return false;
        pl.max_cells = cells > pl.max_cells ? cells : pl.max_cells;
        blk += nb;
        scr += static_cast<long long>(nb) * cells * kCH;
    }
    pl.blk0[L] = blk;
    for (int l = L + 1; l <= kMaxLevels; ++l) pl.blk0[l] = blk;
    pl.scr_per_bm = scr;
    return true;
}

}  // namespace

int64_t msda_tiled_workspace_bytes(const int64_t *shapes_h, const int64_t *start_h, int B, int S, int M, int D, int L, int Lq, int P)
{
    TilePlan pl;
    if (D != kCH || !build_plan(pl, shapes_h, start_h, B, S, M, L, Lq, P)) return 0;
    return 256 + static_cast<int64_t>(B) * M * pl.scr_per_bm * 4;
}

// grad_value must already be zero-filled on `st`.  Returns hipErrorNotSupported when the geometry
// does not qualify (caller falls back to the atomic path).
hipError_t msda_tiled_grad_value_launch(const int64_t *shapes_h, const int64_t *start_h,
                                        const float *loc, const float *attn, const void *grad_out, float *grad_value,
                                        void *workspace, int64_t workspace_bytes,
                                        int B, int S, int M, int D, int L, int Lq, int P, bool absmax_ready, hipStream_t st,
                                        int grad_out_dtype)
{
    if (grad_out_dtype != 0 && !absmax_ready) return hipErrorNotSupported;   // the fp32 absmax pre-pass is not templated
    TilePlan pl;
    if (D != kCH || !build_plan(pl, shapes_h, start_h, B, S, M, L, Lq, P)) return hipErrorNotSupported;
    const int64_t need = 256 + static_cast<int64_t>(B) * M * pl.scr_per_bm * 4;
    if (!workspace || workspace_bytes < need) return hipErrorNotSupported;
    unsigned *absmax2 = static_cast<unsigned *>(workspace);
    float *scratch = reinterpret_cast<float *>(static_cast<unsigned char *>(workspace) + 256);
    hipError_t err = hipSuccess;
    // this path writes its windows where msda_fused.hip keeps a buffer it assumes all-zero between calls: void that claim
    // (bytes 16..31 of the header: cookie + size) so that an A/B run sharing one workspace stays correct
    if ((err = zero_fill_launch(static_cast<unsigned char *>(workspace) + 16, 16, st)) != hipSuccess) return err;
    if (!absmax_ready) {
        if ((err = zero_fill_launch(absmax2, 8, st)) != hipSuccess) return err;
        const int64_t n_go = static_cast<int64_t>(B) * Lq * M * D, n_at = static_cast<int64_t>(B) * Lq * M * L * P;
        hipLaunchKernelGGL(absmax2_kernel, dim3(1024), dim3(256), 0, st, static_cast<const float *>(grad_out), n_go, attn, n_at, absmax2);
    }
    const size_t lds = static_cast<size_t>(pl.max_cells) * kCH * 8 + static_cast<size_t>(kWavesT) * 64 * kRecDwords * 4;
    static bool attr_set[2][64] = {};                        // per kernel instance and device
    const int which = grad_out_dtype == 2 ? 1 : 0;
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = -1;
    if (dev_ < 0 || dev_ >= 64 || !attr_set[which][dev_]) {
        err = which ? hipFuncSetAttribute(reinterpret_cast<const void *>(msda_scatter_tiles<__hip_bfloat16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                    : hipFuncSetAttribute(reinterpret_cast<const void *>(msda_scatter_tiles<float>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (err != hipSuccess) return err;
        if (dev_ >= 0 && dev_ < 64) attr_set[which][dev_] = true;
    }
    const unsigned nblocks = static_cast<unsigned>(B) * M * pl.blk0[L];
    profile_begin(2, Lq, st);
    if (which)
        hipLaunchKernelGGL(msda_scatter_tiles<__hip_bfloat16>, dim3(nblocks), dim3(kThreads), lds, st, pl, loc, attn,
                           static_cast<const __hip_bfloat16 *>(grad_out), grad_value, absmax2, scratch);
    else
        hipLaunchKernelGGL(msda_scatter_tiles<float>, dim3(nblocks), dim3(kThreads), lds, st, pl, loc, attn,
                           static_cast<const float *>(grad_out), grad_value, absmax2, scratch);
    profile_end(st);
    const int64_t nrows = static_cast<int64_t>(B) * S * M;
    profile_begin(3, Lq, st);
    hipLaunchKernelGGL(msda_reduce_tiles, dim3(static_cast<unsigned>((nrows * 8 + 255) / 256)), dim3(256), 0, st, pl, scratch, grad_value);
    profile_end(st);
    return hipGetLastError();
}

}  // namespace mdetr
