// monodetr_amd/csrc/msda_fused.hip -- the whole MSDA backward (reference: ms_deformable_col2im_gpu_kernel_*,
// ms_deform_im2col_cuda.cuh:301-920, and its launcher :956-1326) as ONE kernel per call, D = 32:
// grad_value, grad_sampling_loc and grad_attn_weight from a single pass over loc / attn / grad_out.
//
// What it replaces (round 1: msda.hip `msda_bwd_d32<..,2>` + msda_tiled.hip `msda_scatter_tiles` +
// `msda_reduce_tiles`, 1.14 ms for the encoder call): a gather kernel that produced d/d(loc), d/d(attn), a scatter
// kernel that re-read loc / attn / grad_out and accumulated grad_value in LDS windows WITH A HALO as int64 fixed point
// (one `ds_add_u64` per corner and CHANNEL = 1.34 G LDS atomics, 27 % of its issue slots in the f64 conversion), the
// windows written to a 266 MB scratch buffer and re-read by a reduce kernel: 2.26 GB of HBM traffic for 0.5 GB of
// algorithmic bytes (profiles/r01_msda_pmc.md).
//
// Here:
//   * a workgroup owns (image b, head m, level l, CORE tile of level-l cells) -- cores partition the level, so the
//     window has no halo and is stored straight into grad_value: no scratch round trip, no reduce pass, no zero-fill
//     of grad_value.  Instead of the WINDOW carrying a halo, the set of QUERIES a block looks at does: every query
//     whose pyramid position lies within R cells of the core is a candidate; a candidate's corners that fall into the
//     core are accumulated, the others belong to a neighbouring block (which sees the same query as ITS candidate).
//     Each sample has exactly one OWNER block (the one whose core holds the query's own position) which also gathers
//     the four value rows and produces d/d(loc), d/d(attn) -- the separate gather kernel is gone.
//   * accumulation: TWO channels per 64-bit LDS atomic and no conversion instructions at all.  A contribution
//     x = w * (attn * g * 2^s) is rounded to an integer by ONE fp32 FMA with the addend 1.5 * 2^23 (the integer
//     appears in the low mantissa bits, RNE, |x| < 2^22); the raw IEEE bit patterns of two such results -- exactly what
//     a `v_pk_fma_f32` leaves in a 64-bit register pair -- are added with one `ds_add_u64`.  Integer addition is
//     associative, so the sum is exact and order-independent (deterministic, unlike the reference's fp32 atomics);
//     the constant exponent bits are removed at read-out with the cell's contribution count n (one 32-bit LDS
//     atomic per corner and SAMPLE, not per channel):  sum - n * (C << 32 | C)  leaves  (sum_hi << 32) + sum_lo  in
//     two's complement.  |sum| < 2^31 needs n < 512: the count is checked after the pass and, should a cell have
//     drawn more, the block repeats its pass with contributions pre-scaled by 1/2 (never on the shapes measured).
//   * levels small enough to be held whole (12x40, 6x20: a QUARTER of the pyramid's cells each receives as many samples
//     as level 0) are accumulated per chunk of queries; their partial windows go to a small scratch buffer
//     (39 MB for the encoder call) and a finalize kernel adds them in a fixed order.
//   * a corner that lands outside every block's reach (learned offsets > R cells) is added by its owner with a global
//     fp32 atomic into a side buffer (`far`), which the finalize kernel folds into grad_value only when a device flag
//     says that happened: arbitrary sampling locations stay correct, only slower.  Non-finite grad_out / attn
//     (no scale exists) take that route for every corner, reproducing the reference's NaN / inf propagation.
//   * cross-attention (Lq != S, the decoder: 550 queries): blocks tile the levels the same way and every block scans
//     all queries (2200 samples per level) -- no global atomic at all; the owner of a sample is the tile holding its
//     first corner.
//
// Work unit sizes (B = 8, M = 8, encoder): per (b, m) 15 + 6 core tiles (16 x 32 cells) on levels 0 / 1 and 8 + 8 query
// chunks on levels 2 / 3; a level-0 block looks at ~1 100 queries (510 own) x 4 points.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include <mdetr_wave.h>

#include "msda.h"
#include "mdetr_tune.h"

namespace mdetr {
namespace {

constexpr int kMaxLevels = 4;
constexpr int kCH = 32;               // channels per head (fast path geometry)
constexpr int kRecDw = 12;            // per-sample record, 48 B
constexpr int kCellU64 = kCH / 2;     // packed channel pairs per cell (128 B)
constexpr int kCellStride = kCellU64 + 1;   // 64-bit LDS slots from a cell to the next: one slot of padding rotates consecutive
                                            // cells across the banks (at 128 B every cell would start on bank 0 or 32)
constexpr int kTrash = 8;             // sink rows for corners that are not this block's business (one per record slot mod 8)
constexpr int kMaxCells = 1024;       // (1024 + 8) * 128 B + 4 KB counts + 24 KB records + tables < 160 KB
constexpr unsigned long long kCookie = 0x6d64657472667573ull;

// workspace header (first 256 bytes), followed by the pre-pass's per-workgroup maxima (kMaxPre x 2 words): kHdrBytes in all
constexpr int kMaxPre = 1024;
constexpr int kHdrBytes = 256 + kMaxPre * 2 * 4;
struct Header {
    unsigned npre, pad0;              // workgroups of the pre-pass = valid entries of the maxima array behind the header
    unsigned far;                     // some block added into the `far` buffer during this call
    unsigned pad;
    unsigned long long cookie;        // kCookie once the `far` buffer is known to be all zero between calls ...
    unsigned long long far_elems;     // ... over this many floats (a call of another size lays the workspace out differently)
};

struct FusedPlan {
    int B, S, M, L, P, Lq;
    int H[kMaxLevels], W[kMaxLevels], start[kMaxLevels];
    int mode[kMaxLevels];             // 0: core tiles, candidates = queries within R cells (self-attention over the pyramid)
                                      // 1: whole level resident, queries split into chunks, partial windows -> scratch
                                      // 2: core tiles, every block scans all queries (cross-attention)
                                      // 3: OWNER tiles (self-attention): a block looks at its own queries only and its LDS window
                                      //    carries a halo of R cells; windows are flushed into grad_value with fp32 atomics
    int owner;                        // self-attention in the owner scheme (modes 3 and 1, no candidates, no scratch, no finalize)
    int TH[kMaxLevels], TW[kMaxLevels], nty[kMaxLevels], ntx[kMaxLevels], R[kMaxLevels];
    int nchunk[kMaxLevels];
    // blocks of one (b, m): level l owns blk0[l] + i * bstride[l], i < nblk[l].  Whole-level (chunked) levels with the same
    // chunk count are interleaved (bstride > 1): chunk t of one level and chunk t of the next read the same loc / attn /
    // grad_out lines of the same queries and then run side by side on one XCD, where the second reader hits L2.
    int blk0[kMaxLevels], nblk[kMaxLevels], bstride[kMaxLevels], nblocks;
    long long scr0[kMaxLevels];       // float offset of level l's partial windows within one (b, m) slab (mode 1)
    long long scr_per_bm;
    int max_cells;
    int max_tab;                      // ints of the centre-cell tables a mode-0 block may need
    int npre;                         // workgroups of the pre-pass = pairs of maxima behind the workspace header
    int small;                        // every extent < 2^13: the rectangle bounds and chunk limits fit 32-bit arithmetic
};

// ---- tiny helpers ---------------------------------------------------------------------------------
__host__ __device__ inline int ceil_div_ll(long long a, long long b)   // b > 0, any sign of a
{
    return static_cast<int>(a >= 0 ? (a + b - 1) / b : -((-a) / b));
}

__host__ __device__ inline int row_pad(int tw) { return tw % 16 == 0 ? 1 : 0; }      // tiled windows: see decode_block

// cell of level l (extent n_l) that holds the centre of cell y of a level with extent n_q
__host__ __device__ inline int centre_cell(int y, int n_l, int n_q)
{
    return static_cast<int>((static_cast<unsigned>(2 * y + 1) * static_cast<unsigned>(n_l)) / (2u * static_cast<unsigned>(n_q)));   // extents < 2^15
}

// first y in [0, n_q] with centre_cell(y) >= t   (monotone in y)
__host__ __device__ inline int first_at_or_after(int t, int n_l, int n_q)
{
    const int y = ceil_div_ll(2LL * t * n_q - n_l, 2LL * n_l);
    return y < 0 ? 0 : (y > n_q ? n_q : y);
}

// the same in 32 bits (FusedPlan::small: |t| < 2^15, extents < 2^13): one unsigned division instead of a 64-bit one (~150
// instructions, 33 of them quarter-rate multiplies, per call and per WAVE -- 16 calls per mode-0 block were 10-20 us of its ~35)
__host__ __device__ inline int first_at_or_after_small(int t, int n_l, int n_q)
{
    const int a = 2 * t * n_q - n_l, b = 2 * n_l;
    const int y = a >= 0 ? static_cast<int>((static_cast<unsigned>(a) + static_cast<unsigned>(b) - 1u) / static_cast<unsigned>(b))
                         : -static_cast<int>(static_cast<unsigned>(-a) / static_cast<unsigned>(b));
    return y < 0 ? 0 : (y > n_q ? n_q : y);
}

__device__ __forceinline__ float pix_coord_f(float loc, int size)
{
#pragma clang fp contract(off)
    const float prod = loc * static_cast<float>(size);       // .cuh:285-286: product rounded, then - 0.5
    return prod - 0.5f;
}

__device__ __forceinline__ float sum8f(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    return v;
}

__device__ __forceinline__ float sum4f(float v)            // over the 4 lanes of a quad
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    return v;
}
__device__ __forceinline__ float sum2f(float v)            // over a lane pair
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    return v;
}
template <int LPS> __device__ __forceinline__ float sum_sample(float v) { return LPS == 8 ? sum8f(v) : (LPS == 4 ? sum4f(v) : sum2f(v)); }

// The CPL = 32 / LPS channels ONE lane owns of a 32-channel row (LPS lanes per sample), as they sit in memory: loaded with one
// 16-byte (or 8-byte) request, widened only where fp32 values are needed.
template <typename E, int CPL> struct LaneRaw;
template <> struct LaneRaw<float, 4> {
    typedef f32x4 T;
    static __device__ __forceinline__ T load(const char *p) { return *reinterpret_cast<const f32x4 *>(p); }
    static __device__ __forceinline__ void widen(const T &r, float (&f)[4]) { f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w; }
};
template <> struct LaneRaw<__hip_bfloat16, 4> {
    typedef uint2 T;
    static __device__ __forceinline__ T load(const char *p) { return *reinterpret_cast<const uint2 *>(p); }
    static __device__ __forceinline__ unsigned word(const T &r, int i) { return i == 0 ? r.x : r.y; }
    static __device__ __forceinline__ void widen(const T &r, float (&f)[4])
    {
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xFFFF0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xFFFF0000u);
    }
};
template <> struct LaneRaw<__hip_bfloat16, 8> {
    typedef uint4 T;
    static __device__ __forceinline__ T load(const char *p) { return *reinterpret_cast<const uint4 *>(p); }
    static __device__ __forceinline__ unsigned word(const T &r, int i) { return i == 0 ? r.x : i == 1 ? r.y : i == 2 ? r.z : r.w; }
    static __device__ __forceinline__ void widen(const T &r, float (&f)[8])
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned u = word(r, i);
            f[2 * i] = __uint_as_float(u << 16);
            f[2 * i + 1] = __uint_as_float(u & 0xFFFF0000u);
        }
    }
};

template <> struct LaneRaw<__hip_bfloat16, 16> {             // half a row per lane (LPS = 2): two 16-byte requests
    struct T { uint4 a, b; };
    static __device__ __forceinline__ T load(const char *p)
    {
        T r;
        r.a = *reinterpret_cast<const uint4 *>(p);
        r.b = *reinterpret_cast<const uint4 *>(p + 16);
        return r;
    }
    static __device__ __forceinline__ unsigned word(const T &r, int i)
    {
        return i == 0 ? r.a.x : i == 1 ? r.a.y : i == 2 ? r.a.z : i == 3 ? r.a.w : i == 4 ? r.b.x : i == 5 ? r.b.y : i == 6 ? r.b.z : r.b.w;
    }
    static __device__ __forceinline__ void widen(const T &r, float (&f)[16])
    {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned u = word(r, i);
            f[2 * i] = __uint_as_float(u << 16);
            f[2 * i + 1] = __uint_as_float(u & 0xFFFF0000u);
        }
    }
};

// the four corner dot products e[c] = sum_i g[i] * v_c[i] over this lane's channels
template <typename GT, typename VT, int CPL> struct CornerDots {
    // packed pairs: (e0, e1) and (e2, e3) share their FMAs
    static __device__ __forceinline__ void run(const typename LaneRaw<GT, CPL>::T &, const float (&g)[CPL],
                                               const typename LaneRaw<VT, CPL>::T (&v)[4], float (&e)[4])
    {
        float f0[CPL], f1[CPL], f2[CPL], f3[CPL];            // (four separate arrays: a 2-D one is not promoted to registers)
        LaneRaw<VT, CPL>::widen(v[0], f0); LaneRaw<VT, CPL>::widen(v[1], f1);
        LaneRaw<VT, CPL>::widen(v[2], f2); LaneRaw<VT, CPL>::widen(v[3], f3);
        f32x2 p01 = mul2(make_f32x2(g[0], g[0]), make_f32x2(f0[0], f1[0])), p23 = mul2(make_f32x2(g[0], g[0]), make_f32x2(f2[0], f3[0]));
#pragma unroll
        for (int i = 1; i < CPL; ++i) {
            p01 = fma2(make_f32x2(g[i], g[i]), make_f32x2(f0[i], f1[i]), p01);
            p23 = fma2(make_f32x2(g[i], g[i]), make_f32x2(f2[i], f3[i]), p23);
        }
        e[0] = p01.x; e[1] = p01.y; e[2] = p23.x; e[3] = p23.y;
    }
};
template <int CPL> struct CornerDots<__hip_bfloat16, __hip_bfloat16, CPL> {
    // both operands bf16: v_dot2c_f32_bf16 on the words as loaded (products exact in fp32, fp32 accumulation) -- no widening
    static __device__ __forceinline__ void run(const typename LaneRaw<__hip_bfloat16, CPL>::T &graw, const float (&)[CPL],
                                               const typename LaneRaw<__hip_bfloat16, CPL>::T (&v)[4], float (&e)[4])
    {
        typedef LaneRaw<__hip_bfloat16, CPL> R;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < CPL / 2; ++i) acc = dot2_bf16(R::word(graw, i), R::word(v[c], i), acc);
            e[c] = acc;
        }
    }
};

// ---- 1. scale pre-pass: max|grad_out|, max|attn| (and the first-use zero fill of the `far` buffer) ---------------
template <typename GT>
__global__ __launch_bounds__(256)
void msda_absmax_kernel(const GT *__restrict__ g, int64_t ng, const float *__restrict__ a, int64_t na,
                        Header *__restrict__ hdr, float *__restrict__ far, int64_t nfar, int zero_always)
{
    // ng, na are multiples of 4; bases 16-byte aligned
    float mg = 0.f, ma = 0.f, poison = 0.f;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t tid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    auto take = [&](const float4 &v, float &mx) __attribute__((always_inline)) {
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        poison += (v.x + v.y + v.z + v.w) * 0.f;             // NaN, or inf * 0, poisons the sum
    };
    // four independent 16-byte requests in flight per lane (a pure HBM stream: 84 MB for the encoder call)
    if constexpr (sizeof(GT) == 2) {
        // bf16: 16 bytes = 8 values per request (4-value requests are 8 bytes: half the bytes per instruction, and the 42 MB of
        // grad_out took 28 of this pass's 36 us).  |x| of a bf16 is its low 15 bits, non-negative floats order like their bit
        // patterns: the running maximum is an integer maximum of the masked halves; inf / NaN (exponent all ones) come out >= 0x7F80.
        const uint4 *gp = reinterpret_cast<const uint4 *>(g);
        const int64_t nv = ng / 8;
        unsigned mlo = 0u, mhi = 0u;                          // maxima of the even / odd elements (low / high halves)
        auto take8 = [&](const uint4 &v) __attribute__((always_inline)) {
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned lo = w[k] & 0x7FFFu, hi = (w[k] >> 16) & 0x7FFFu;
                mlo = lo > mlo ? lo : mlo;
                mhi = hi > mhi ? hi : mhi;
            }
        };
        int64_t i = tid;
        for (; i + 3 * stride < nv; i += 4 * stride) {
            const uint4 v0 = gp[i], v1 = gp[i + stride], v2 = gp[i + 2 * stride], v3 = gp[i + 3 * stride];
            take8(v0); take8(v1); take8(v2); take8(v3);
        }
        for (; i < nv; i += stride) take8(gp[i]);
        if (tid == 0)                                         // (ng is a multiple of 4: at most one 4-value tail)
            for (int64_t t = nv * 8; t < ng; ++t) {
                const unsigned b = reinterpret_cast<const unsigned short *>(g)[t] & 0x7FFFu;
                mlo = b > mlo ? b : mlo;
            }
        const unsigned m16 = mlo > mhi ? mlo : mhi;
        mg = m16 >= 0x7F80u ? __builtin_inff() : __uint_as_float(m16 << 16);
    } else {
        const char *gp = reinterpret_cast<const char *>(g);
        constexpr int64_t eb4 = 4 * Elem<GT>::kBytes;
        int64_t i = tid;
        for (; i + 3 * stride < ng / 4; i += 4 * stride) {
            const float4 v0 = Elem<GT>::load4(gp + i * eb4), v1 = Elem<GT>::load4(gp + (i + stride) * eb4);
            const float4 v2 = Elem<GT>::load4(gp + (i + 2 * stride) * eb4), v3 = Elem<GT>::load4(gp + (i + 3 * stride) * eb4);
            take(v0, mg); take(v1, mg); take(v2, mg); take(v3, mg);
        }
        for (; i < ng / 4; i += stride) take(Elem<GT>::load4(gp + i * eb4), mg);
    }
    {
        const float4 *ap = reinterpret_cast<const float4 *>(a);
        int64_t i = tid;
        for (; i + 3 * stride < na / 4; i += 4 * stride) {
            const float4 v0 = ap[i], v1 = ap[i + stride], v2 = ap[i + 2 * stride], v3 = ap[i + 3 * stride];
            take(v0, ma); take(v1, ma); take(v2, ma); take(v3, ma);
        }
        for (; i < na / 4; i += stride) take(ap[i], ma);
    }
    if (!(poison == 0.f)) mg = __builtin_inff();
    for (int o = 32; o > 0; o >>= 1) { mg = fmaxf(mg, __shfl_xor(mg, o)); ma = fmaxf(ma, __shfl_xor(ma, o)); }
    // one pair of words per WORKGROUP behind the header, reduced by every block of the main kernel (1 024 x 2 words from L2): no
    // atomics at all.  (Same-address atomicMax serialises in L2, ~12 ns each: one per workgroup made this 20-microsecond pass
    // take 35, one per wave 110; and the words needed a zero-fill launch of their own in front of the pass.)
    __shared__ unsigned s_max[2];
    if (threadIdx.x < 2) s_max[threadIdx.x] = 0u;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {                            // non-negative floats order like their bit patterns
        atomicMax(&s_max[0], __builtin_bit_cast(unsigned, mg));
        atomicMax(&s_max[1], __builtin_bit_cast(unsigned, ma));
    }
    __syncthreads();
    unsigned *maxima = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(hdr) + 256);
    if (threadIdx.x < 2) maxima[2 * blockIdx.x + threadIdx.x] = s_max[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hdr->npre = gridDim.x;
        hdr->far = 0u;                                        // (set by the main kernel, read by the finalize pass: stream order)
    }
    // a workspace this library has not finalized yet (fresh allocation): its `far` buffer may hold anything
    // (owner scheme: `far` IS grad_value, which every block of the main kernel adds into: zero-filled by this pass, every call)
    if (zero_always || hdr->cookie != kCookie || hdr->far_elems != static_cast<unsigned long long>(nfar)) {
        float4 *f4 = reinterpret_cast<float4 *>(far);
        for (int64_t i = tid; i < nfar / 4; i += stride) f4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ---- 2. the fused backward -----------------------------------------------------------------------------------------
struct Work {
    int l, mode;
    int cy0, cx0;                      // core origin in level-l cells
    int tstride;                       // cells per window row
    int ncell;                         // window cells (stride-based)
    // mode 0: the candidate queries are one rectangle per query level, enumerated level by level, row by row
    int y0[kMaxLevels], x0[kMaxLevels], wx[kMaxLevels], rows[kMaxLevels], cum[kMaxLevels + 1];
    float inv_wx[kMaxLevels];
    int roff[kMaxLevels], coff[kMaxLevels], ntab;         // centre-cell tables (LDS): rows / columns of each rectangle
    int q0, nq;                        // number of candidate queries (mode 1 / 2: contiguous from q0)
    int slot;                          // tile / chunk index within the level
};

// `bounds` = 16 ints of LDS scratch (tiled levels of self-attention): the 4 x 4 rectangle bounds of the block are computed ONCE,
// one per lane of the first 16 threads, and read back by everyone -- not 16 serial divisions on each of the block's waves.
// Called by every thread of the block (two barriers inside, the branch is block-uniform).
__device__ __forceinline__ Work decode_block(const FusedPlan &pl, int k, int *bounds)
{
    Work w;
    int l = 0;
#pragma unroll
    for (int i = 0; i < kMaxLevels; ++i) {
        if (i < pl.L && k >= pl.blk0[i]) {
            const unsigned d = static_cast<unsigned>(k - pl.blk0[i]), st = static_cast<unsigned>(pl.bstride[i]);
            const unsigned qd = st == 1u ? d : d / st;
            if (qd * st == d && qd < static_cast<unsigned>(pl.nblk[i])) l = i;
        }
    }
    w.l = l;
    w.mode = pl.mode[l];
    const unsigned dk = static_cast<unsigned>(k - pl.blk0[l]), stl = static_cast<unsigned>(pl.bstride[l]);
    const int t = static_cast<int>(stl == 1u ? dk : dk / stl);
    w.slot = t;
    w.q0 = 0;
    w.nq = 0;
    w.ntab = 0;
    w.cum[0] = 0;
#pragma unroll
    for (int lq = 0; lq < kMaxLevels; ++lq) {
        w.y0[lq] = w.x0[lq] = w.rows[lq] = w.roff[lq] = w.coff[lq] = 0;
        w.wx[lq] = 1; w.inv_wx[lq] = 1.f; w.cum[lq + 1] = 0;
    }
    if (w.mode == 1) {
        w.cy0 = 0; w.cx0 = 0; w.tstride = pl.W[l]; w.ncell = pl.H[l] * pl.W[l];
        const int nch = pl.nchunk[l];
        if (static_cast<long long>(pl.Lq) * (nch + 1) < (1LL << 31)) {          // (uniform) the chunk limits in 32 bits
            const unsigned un = static_cast<unsigned>(nch), lq_ = static_cast<unsigned>(pl.Lq);
            w.q0 = static_cast<int>(lq_ * static_cast<unsigned>(t) / un);
            w.nq = static_cast<int>(lq_ * static_cast<unsigned>(t + 1) / un) - w.q0;
        } else {
            w.q0 = static_cast<int>(static_cast<long long>(pl.Lq) * t / nch);
            w.nq = static_cast<int>(static_cast<long long>(pl.Lq) * (t + 1) / nch) - w.q0;
        }
    } else {
        const unsigned ntx = static_cast<unsigned>(pl.ntx[l]);
        const int ty = static_cast<int>(static_cast<unsigned>(t) / ntx), tx = t - ty * static_cast<int>(ntx);
        // (a window row of a multiple of 16 cells gets one cell of padding: with 136-byte cells, rows 32 cells apart start on the
        // same LDS bank -- the four points of a query of a head whose offsets run along y then collide 4-way in every ds_add_u64;
        // that was the 37 % bank-conflict share of the LDS cycles in rounds 2 and 3)
        w.cy0 = ty * pl.TH[l]; w.cx0 = tx * pl.TW[l]; w.tstride = pl.TW[l] + row_pad(pl.TW[l]); w.ncell = pl.TH[l] * w.tstride;
        if (w.mode == 2) {
            w.nq = pl.Lq;
        } else {
            // mode 3 (owner): own queries only -- on every query level the rectangle of cells whose centre falls into the CORE
            // (the window adds R cells); mode 0: the queries whose centre lies within R cells of the core
            const int R = pl.R[l], margin = w.mode == 3 ? 0 : R;
            if (threadIdx.x < 4 * kMaxLevels) {
                const int lq = static_cast<int>(threadIdx.x) >> 2, which = static_cast<int>(threadIdx.x) & 3;
                int v = 0;
                if (lq < pl.L) {
                    const bool xs = (which & 2) != 0, hi = (which & 1) != 0;
                    const int t0 = (xs ? w.cx0 : w.cy0) + (hi ? (xs ? pl.TW[l] : pl.TH[l]) + margin : -margin);
                    const int n_l = xs ? pl.W[l] : pl.H[l];
                    // (pl.H[lq] / pl.W[lq] with a lane-dependent index: selected, not indexed -- the plan sits in SGPRs)
                    const int hq = lq == 0 ? pl.H[0] : lq == 1 ? pl.H[1] : lq == 2 ? pl.H[2] : pl.H[3];
                    const int wq = lq == 0 ? pl.W[0] : lq == 1 ? pl.W[1] : lq == 2 ? pl.W[2] : pl.W[3];
                    const int n_q = xs ? wq : hq;
                    v = pl.small ? first_at_or_after_small(t0, n_l, n_q) : first_at_or_after(t0, n_l, n_q);
                }
                bounds[threadIdx.x] = v;                      // [lq][y lo, y hi, x lo, x hi]
            }
            __syncthreads();
            int bnd[4 * kMaxLevels];
#pragma unroll
            for (int i = 0; i < 4 * kMaxLevels; ++i) bnd[i] = __builtin_amdgcn_readfirstlane(bounds[i]);
            __syncthreads();                                  // the scratch is reused by the caller
#pragma unroll
            for (int lq = 0; lq < kMaxLevels; ++lq) {
                if (lq < pl.L) {
                    w.y0[lq] = bnd[4 * lq];
                    w.rows[lq] = bnd[4 * lq + 1] - bnd[4 * lq];
                    w.x0[lq] = bnd[4 * lq + 2];
                    const int cols = bnd[4 * lq + 3] - bnd[4 * lq + 2];
                    w.wx[lq] = cols > 0 ? cols : 1;
                    w.inv_wx[lq] = 1.0f / static_cast<float>(w.wx[lq]);
                    if (w.mode == 3) {
                        w.cum[lq + 1] = w.cum[lq] + (cols > 0 ? w.rows[lq] * cols : 0);
                    } else {
                        w.roff[lq] = w.ntab; w.ntab += w.rows[lq];
                        w.coff[lq] = w.ntab; w.ntab += cols;
                        w.cum[lq + 1] = w.cum[lq] + w.rows[lq] * cols;
                    }
                } else {
                    w.cum[lq + 1] = w.cum[lq];
                }
            }
            w.nq = w.cum[kMaxLevels];
            if (w.mode == 3) {
                w.cy0 -= R; w.cx0 -= R;                       // from here on: the WINDOW's origin and extent
                w.tstride = pl.TW[l] + 2 * R;
                w.ncell = (pl.TH[l] + 2 * R) * w.tstride;
            }
        }
    }
    return w;
}

// four corner contributions of one sample for the CPL channels of this lane: per corner CPL / 2 packed FMAs and as many
// ds_add_u64 (the 64-bit register pair a packed FMA leaves behind IS the atomic's operand)
template <int CPL>
__device__ __forceinline__ void accumulate(unsigned long long *win, unsigned o01, unsigned o23, const float (&wt)[4],
                                           const float (&ag)[CPL], int k, float magic)
{
    unsigned cell[4] = {o01 & 0xFFFFu, o01 >> 16, o23 & 0xFFFFu, o23 >> 16};
    const f32x2 mg = make_f32x2(magic, magic);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        unsigned long long *p = win + cell[c] * kCellStride + (CPL / 2) * k;
        const f32x2 wc = make_f32x2(wt[c], wt[c]);
#pragma unroll
        for (int pr = 0; pr < CPL / 2; ++pr) {
            const f32x2 r = fma2(wc, make_f32x2(ag[2 * pr], ag[2 * pr + 1]), mg);
            __hip_atomic_fetch_add(p + pr, __builtin_bit_cast(unsigned long long, r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// LDS carve-up (bytes), shared by the kernel and the launcher
__host__ __device__ inline size_t lds_win_bytes(int max_cells) { return (static_cast<size_t>(max_cells + kTrash) * kCellStride * 8 + 15) & ~static_cast<size_t>(15); }
__host__ __device__ inline size_t lds_cnt_bytes(int max_cells) { return (static_cast<size_t>(max_cells) * 4 + 15) & ~static_cast<size_t>(15); }

// LPS lanes per sample (each owning CPL = 32 / LPS channels), GMAX groups of 64 / LPS own samples with their loads in flight
// together.  bf16: LPS = 4 -- 16 samples per wave pass instead of 8 halves every per-sample instruction (record reads, cell
// addresses, the lane reduction, the d/d(loc) arithmetic) and the dot products run on the packed words.
template <typename VT, typename GT, int THREADS, int GMAX, int LPS, bool OWNER = false>
__global__ __launch_bounds__(THREADS, (THREADS == 512 && GMAX <= 4 && sizeof(VT) == 2) ? 4 : 2)
void msda_bwd_fused(const FusedPlan pl, const VT *__restrict__ value, const float *__restrict__ loc,
                    const float *__restrict__ attn, const GT *__restrict__ grad_out,
                    float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_attn,
                    Header *__restrict__ hdr, float *__restrict__ scratch, float *__restrict__ far)
{
    constexpr int kWavesB = THREADS / 64;
    constexpr int CPL = kCH / LPS, SPG = 64 / LPS;            // channels per lane, samples per group (one wave pass)
    typedef LaneRaw<VT, CPL> RV;
    typedef LaneRaw<GT, CPL> RG;
    MDETR_DYNAMIC_LDS(unsigned char, smem_raw);
    unsigned long long *win = reinterpret_cast<unsigned long long *>(smem_raw);
    unsigned *cnt = reinterpret_cast<unsigned *>(smem_raw + lds_win_bytes(pl.max_cells));
    unsigned *recs = reinterpret_cast<unsigned *>(smem_raw + lds_win_bytes(pl.max_cells) + lds_cnt_bytes(pl.max_cells));
    int *tab = reinterpret_cast<int *>(recs + kWavesB * 64 * kRecDw);       // centre cells of the candidate rows / columns (mode 0)
    int *lv = tab + pl.max_tab;                                              // 4 x 8 ints: the candidate rectangle of each query level
    unsigned *blk = reinterpret_cast<unsigned *>(lv + 32);                   // [0] = max count of the pass

    const int bid = blockIdx.x;
    const int b = bid % pl.B, r_ = bid / pl.B, m = r_ % pl.M, kblk = r_ / pl.M;      // image -> XCD (bid % 8)
    unsigned *wmax = blk + 4;                                                // 2 words per wave: the pre-pass's maxima, reduced
    // the pre-pass left one pair of maxima per workgroup behind the header: requested first, consumed after the block's bounds
    unsigned mg_ = 0u, ma_ = 0u;
    {
        const uint2 *maxima = reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned char *>(hdr) + 256);
        for (int i = threadIdx.x; i < pl.npre; i += THREADS) {
            const uint2 v = maxima[i];
            mg_ = v.x > mg_ ? v.x : mg_;
            ma_ = v.y > ma_ ? v.y : ma_;
        }
    }
    const Work w = decode_block(pl, kblk, lv);
    // TH x TW: the cells this block accumulates in LDS -- the core tile, the whole level (mode 1), or the core plus its halo (mode 3)
    const int l = w.l, H = pl.H[l], W = pl.W[l];
    const int TH = w.mode == 1 ? H : pl.TH[l] + (w.mode == 3 ? 2 * pl.R[l] : 0), TW = w.mode == 1 ? W : pl.TW[l] + (w.mode == 3 ? 2 * pl.R[l] : 0);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int P = pl.P, LP = pl.L * P, M = pl.M, R = pl.R[l];
    const int j = lane / LPS, k = lane % LPS;
    constexpr int eb = Elem<VT>::kBytes;
    const int rowb = M * kCH * eb;                            // bytes from a pixel to the next, `value`
    const char *vlev = reinterpret_cast<const char *>(value) + ((static_cast<int64_t>(b) * pl.S + pl.start[l]) * M + m) * (kCH * eb) + k * CPL * eb;
    const char *gbase = reinterpret_cast<const char *>(grad_out) + k * CPL * Elem<GT>::kBytes;
    // index arithmetic in 32 bits (the plan guarantees B * Lq * M * L * P < 2^28 and byte offsets into grad_out / one image's
    // value below 2^31): pair index of query q = pair0u + q * M
    const unsigned pair0u = static_cast<unsigned>(b) * static_cast<unsigned>(pl.Lq) * M + m;
    // corners beyond the block's cells: global fp32 atomics -- into the side buffer the finalize pass folds in, or (owner scheme:
    // grad_value is zero-filled by the pre-pass and every block adds into it) straight into grad_value
    float *far_lev = (OWNER ? grad_value : far) + ((static_cast<int64_t>(b) * pl.S + pl.start[l]) * M + m) * kCH + k * CPL;
    unsigned *wrec = recs + wave * 64 * kRecDw;

    // centre cells of the candidate rectangles' rows and columns on level l (mode 0): two small tables instead of two
    // divisions per candidate
    if (w.mode == 0 || w.mode == 3) {
        if (threadIdx.x < kMaxLevels) {
            const int lq = threadIdx.x;
#define MDETR_SEL4(arr) (lq == 0 ? arr[0] : lq == 1 ? arr[1] : lq == 2 ? arr[2] : arr[3])
            int *L8 = lv + lq * 8;
            L8[0] = MDETR_SEL4(w.cum); L8[1] = MDETR_SEL4(w.wx); L8[2] = __builtin_bit_cast(int, MDETR_SEL4(w.inv_wx));
            L8[3] = MDETR_SEL4(w.y0); L8[4] = MDETR_SEL4(w.x0); L8[5] = MDETR_SEL4(pl.W); L8[6] = MDETR_SEL4(pl.start);
            L8[7] = MDETR_SEL4(w.roff) | (MDETR_SEL4(w.coff) << 16);
#undef MDETR_SEL4
        }
        for (int t = threadIdx.x; w.mode == 0 && t < w.ntab; t += THREADS) {
            int v = 0;
#pragma unroll
            for (int lq = 0; lq < kMaxLevels; ++lq) {
                if (lq < pl.L) {
                    if (t >= w.roff[lq] && t < w.coff[lq]) v = centre_cell(w.y0[lq] + t - w.roff[lq], H, pl.H[lq]);
                    if (t >= w.coff[lq] && t < w.coff[lq] + w.wx[lq]) v = centre_cell(w.x0[lq] + t - w.coff[lq], W, pl.W[lq]);
                }
            }
            tab[t] = v;
        }
    }

    // one power-of-two scale per call: |w * attn * g| <= max|attn| * max|g| = mx < 2^e (every block reduces the pre-pass's per-
    // workgroup maxima: one 8-byte load per thread above, a wave reduction, 2 words of LDS per wave).  The same barrier covers the
    // first zero fill of the window.
    for (int i = threadIdx.x; i < (w.ncell + kTrash) * kCellStride; i += THREADS) win[i] = 0ull;
    for (int i = threadIdx.x; i < w.ncell; i += THREADS) cnt[i] = 0u;
    if (threadIdx.x == 0) { blk[0] = 0u; blk[1] = kWavesB * 64u; blk[2] = 0u; blk[3] = 0u; }
    float mx;
    {
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned og = __shfl_xor(mg_, o), oa = __shfl_xor(ma_, o);
            mg_ = og > mg_ ? og : mg_;
            ma_ = oa > ma_ ? oa : ma_;
        }
        if (lane == 0) { wmax[2 * wave] = mg_; wmax[2 * wave + 1] = ma_; }
        __syncthreads();
        unsigned mg = 0u, ma = 0u;
#pragma unroll
        for (int i = 0; i < kWavesB; ++i) {
            const unsigned vg = wmax[2 * i], va = wmax[2 * i + 1];
            mg = vg > mg ? vg : mg;
            ma = va > ma ? va : ma;
        }
        mx = __builtin_bit_cast(float, mg) * __builtin_bit_cast(float, ma);
    }
    const bool finite = mx <= 3.0e38f;                        // inf / NaN somewhere: every corner goes to the `far` buffer
    int e = 0;
    if (finite && mx > 0.f) (void)frexpf(mx, &e);
    const int nsamp = w.nq * P;
    const float inv_p = 1.0f / static_cast<float>(P);

    for (int shift = 0;; ++shift) {
        // contributions are rounded to multiples of 2^-(22 - shift - e): |x| * scale < 2^(22 - shift), up to 2^(9 + shift) - 1 per cell
        const float scale = ldexpf(1.0f, 22 - shift - e);
        const float magic = ldexpf(1.0f, 23) + ldexpf(1.0f, 22 - shift);
        if (shift > 0) {                                      // (the first pass's zero fill shares the prologue's barrier)
            for (int i = threadIdx.x; i < (w.ncell + kTrash) * kCellStride; i += THREADS) win[i] = 0ull;
            for (int i = threadIdx.x; i < w.ncell; i += THREADS) cnt[i] = 0u;
            if (threadIdx.x == 0) { blk[0] = 0u; blk[1] = kWavesB * 64u; blk[2 + (shift & 1)] = 0u; }
            __syncthreads();
        }

        // candidate decode + the loads of its location / weight, issued one step AHEAD of their use (the loop is bound by
        // memory round trips, not by arithmetic: everything that can be in flight early is)
        struct Cand { int q, p, qcy, qcx; bool act, owned; float2 xy; float a; };
        auto decode = [&](int base_, Cand &c) {
            const int i = base_ + lane;
            c.act = i < nsamp;
            const int ia = c.act ? i : 0;
            // (exact for these sizes: the quotient is < 2^20 and sits 0.5 / P away from the next integer)
            const int qi = P == 4 ? ia >> 2 : static_cast<int>((static_cast<float>(ia) + 0.5f) * inv_p);
            c.p = ia - qi * P;
            c.qcy = c.qcx = 0;
            if (w.mode == 0 || w.mode == 3) {
                const int lq = (qi >= w.cum[1] ? 1 : 0) + (qi >= w.cum[2] ? 1 : 0) + (qi >= w.cum[3] ? 1 : 0);
                const int *L8 = lv + lq * 8;                  // this query level's rectangle (LDS; a step rarely straddles two levels)
                const int rel = qi - L8[0], wxl = L8[1], y0l = L8[3], x0l = L8[4], Wq = L8[5], sq = L8[6], ro = L8[7] & 0xFFFF, co = L8[7] >> 16;
                const float inv = __builtin_bit_cast(float, L8[2]);
                const int row = static_cast<int>((static_cast<float>(rel) + 0.5f) * inv), col = rel - row * wxl;
                c.q = sq + (y0l + row) * Wq + x0l + col;
                if (!OWNER) {
                    c.qcy = tab[ro + row];
                    c.qcx = tab[co + col];
                    c.owned = static_cast<unsigned>(c.qcy - w.cy0) < static_cast<unsigned>(TH) && static_cast<unsigned>(c.qcx - w.cx0) < static_cast<unsigned>(TW);
                } else {
                    c.owned = true;                           // the rectangles hold exactly the queries whose centre lies in the core
                }
            } else {
                c.q = w.q0 + qi;
                c.owned = w.mode == 1;
            }
            const unsigned srec = (pair0u + static_cast<unsigned>(c.q) * M) * LP + l * P + c.p;
            c.xy = c.act ? *reinterpret_cast<const float2 *>(loc + srec * 2) : make_float2(-9.f, -9.f);
            c.a = c.act ? attn[srec] : 0.f;
        };

        // steps of 64 candidates: the first one per wave by position, the following ones from a counter in LDS -- the steps differ
        // (a row of the candidate rectangle inside the core is all own samples, a row of its margin none), and with a fixed
        // stride the waves of a block finished up to a third apart (the barrier behind the loop was 17 % of a block's time)
        Cand cur;
        if (wave * 64 < nsamp) decode(wave * 64, cur);
        for (int base = wave * 64; base < nsamp;) {
            unsigned nxt_ = 0u;
            if (lane == 0) nxt_ = atomicAdd(blk + 1, 64u);
            const int next_base = __builtin_amdgcn_readfirstlane(static_cast<int>(nxt_));
            // ---- a. one candidate sample per lane: footprint, ownership, record ----------------------------------------
            bool keep = false, owned = false;
            unsigned rec[kRecDw];
            {
                const bool act = cur.act;
                const int q = cur.q, p = cur.p, qcy = cur.qcy, qcx = cur.qcx;
                owned = cur.owned;
                const float2 xy = cur.xy;
                const float a_in = cur.a;
                const float h_im = pix_coord_f(xy.y, H), w_im = pix_coord_f(xy.x, W);
                const bool inwin = h_im > -1.f && w_im > -1.f && h_im < static_cast<float>(H) && w_im < static_cast<float>(W);   // .cuh:288
                const float hs = inwin ? h_im : 0.f, ws = inwin ? w_im : 0.f;
                const float hf = floorf(hs), wf = floorf(ws);
                const int y = static_cast<int>(hf), x = static_cast<int>(wf);
                const float lh = hs - hf, lw = ws - wf, hh = 1.f - lh, hw = 1.f - lw;
                const float cw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                const int yc0 = min(max(y, 0), H - 1), yc1 = min(max(y + 1, 0), H - 1);
                const int xc0 = min(max(x, 0), W - 1), xc1 = min(max(x + 1, 0), W - 1);
                if (w.mode == 2) owned = static_cast<unsigned>(yc0 - w.cy0) < static_cast<unsigned>(TH) && static_cast<unsigned>(xc0 - w.cx0) < static_cast<unsigned>(TW);
                owned = owned && act;
                // per axis: inside the map (.cuh:56-74), inside this block's core (one unsigned compare each)
                const int wy0 = y - w.cy0, wx0 = x - w.cx0;
                const bool aw = act && inwin;
                const bool my[2] = {aw && static_cast<unsigned>(y) < static_cast<unsigned>(H), aw && static_cast<unsigned>(y + 1) < static_cast<unsigned>(H)};
                const bool mx[2] = {static_cast<unsigned>(x) < static_cast<unsigned>(W), static_cast<unsigned>(x + 1) < static_cast<unsigned>(W)};
                const bool ky[2] = {finite && static_cast<unsigned>(wy0) < static_cast<unsigned>(TH), finite && static_cast<unsigned>(wy0 + 1) < static_cast<unsigned>(TH)};
                const bool kx[2] = {static_cast<unsigned>(wx0) < static_cast<unsigned>(TW), static_cast<unsigned>(wx0 + 1) < static_cast<unsigned>(TW)};
                const int cell0 = wy0 * w.tstride + wx0;
                unsigned cell[4], valid = 0u, stray = 0u;
                float wt[4];
                bool anycore = false;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool inmap = my[c >> 1] && mx[c & 1];
                    wt[c] = inmap ? cw[c] : 0.f;
                    valid |= inmap ? (1u << c) : 0u;
                    // grad_value: a corner of bilinear weight exactly 0 (integer pixel coordinates, as the model's initial
                    // offsets produce) adds exactly 0 for finite gradients -- it is neither accumulated nor counted
                    const bool ok = inmap && (wt[c] != 0.f || !finite);
                    const bool core = ok && ky[c >> 1] && kx[c & 1];
                    cell[c] = core ? static_cast<unsigned>(cell0 + (c >> 1) * w.tstride + (c & 1)) : 0xFFFFu;
                    anycore = anycore || core;
                    stray |= (ok && !core && owned) ? (1u << c) : 0u;
                }
                // an own sample's corner outside the core: does the block that owns its cell look at this query?  That block is
                // this one's neighbour in the direction of the corner (a corner a whole tile further out is beyond anybody's
                // reach: R <= TH, TW); it looks at queries whose centre lies within R cells of ITS core.  (Rare: skipped
                // altogether when no lane of the wave has such a corner.)
                unsigned farm = stray;                        // (owner scheme: a corner beyond the window's halo goes out as a global atomic)
                if (finite && w.mode == 2) farm = 0u;         // every other block scans every query
                if (!OWNER && finite && w.mode == 0 && __any(stray != 0u)) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int wy = y + (c >> 1) - w.cy0, wx = x + (c & 1) - w.cx0;
                        const int oy = w.cy0 + (wy < 0 ? -TH : (wy >= TH ? TH : 0)), ox = w.cx0 + (wx < 0 ? -TW : (wx >= TW ? TW : 0));
                        const bool beyond = wy < -TH || wy >= 2 * TH || wx < -TW || wx >= 2 * TW;
                        const bool other = !beyond && qcy >= oy - R && qcy < oy + TH + R && qcx >= ox - R && qcx < ox + TW + R;
                        farm &= other ? ~(1u << c) : ~0u;
                    }
                }
                keep = owned || anycore;
                rec[0] = static_cast<unsigned>(q);
                rec[1] = static_cast<unsigned>(yc0 * W + xc0) | (static_cast<unsigned>(xc1 - xc0) << 30) | (static_cast<unsigned>(yc1 - yc0) << 31);
                rec[2] = cell[0] | (cell[1] << 16);
                rec[3] = cell[2] | (cell[3] << 16);
#pragma unroll
                for (int c = 0; c < 4; ++c) rec[4 + c] = __builtin_bit_cast(unsigned, wt[c]);
                rec[8] = __builtin_bit_cast(unsigned, inwin ? a_in : 0.f);
                rec[9] = __builtin_bit_cast(unsigned, lh);
                rec[10] = __builtin_bit_cast(unsigned, lw);
                rec[11] = valid | (farm << 4) | (static_cast<unsigned>(p) << 8);
            }
            if (next_base < nsamp) decode(next_base, cur);                              // next step's loads go out now
            // own samples are listed from the front of the wave's record buffer, neighbours' from the back (owner scheme: every
            // active lane is an own sample and the active lanes are a prefix of the wave: the list is the lanes themselves)
            const unsigned long long mo = OWNER ? 0ull : __ballot(keep && owned), mh = OWNER ? 0ull : __ballot(keep && !owned);
            const int no = OWNER ? min(64, nsamp - base) : __popcll(mo), nh = OWNER ? 0 : __popcll(mh);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (keep) {
                const int slot = OWNER ? lane : (owned ? __popcll(mo & below) : 63 - __popcll(mh & below));
#pragma unroll
                for (int c = 0; c < 4; ++c) {                 // this block's cells: count the contribution, other corners -> sink row
                    const unsigned cc = (c < 2 ? rec[2] >> (16 * c) : rec[3] >> (16 * (c - 2))) & 0xFFFFu;
                    if (cc != 0xFFFFu) atomicAdd(cnt + cc, 1u);
                }
                const unsigned sink = static_cast<unsigned>(w.ncell + (slot & 7));
                const unsigned c0 = rec[2] & 0xFFFFu, c1 = rec[2] >> 16, c2 = rec[3] & 0xFFFFu, c3 = rec[3] >> 16;
                rec[2] = (c0 == 0xFFFFu ? sink : c0) | ((c1 == 0xFFFFu ? sink : c1) << 16);
                rec[3] = (c2 == 0xFFFFu ? sink : c2) | ((c3 == 0xFFFFu ? sink : c3) << 16);
                unsigned *dst = wrec + slot * kRecDw;
                *reinterpret_cast<uint4 *>(dst) = make_uint4(rec[0], rec[1], rec[2], rec[3]);
                *reinterpret_cast<uint4 *>(dst + 4) = make_uint4(rec[4], rec[5], rec[6], rec[7]);
                *reinterpret_cast<uint4 *>(dst + 8) = make_uint4(rec[8], rec[9], rec[10], rec[11]);
            }
            wave_sync();

            // ---- b1. own samples, SPG per group (LPS lanes x CPL channels each), up to GMAX groups per batch: every load of the
            //          batch is requested before the first group is consumed ---------------------------------------------------
            auto run_own = [&](bool on, int r, const typename RG::T &graw, const typename RV::T (&vraw)[4]) __attribute__((always_inline)) {
                // (lanes beyond the list hold the list's last record and its loads: everything is computed, only the side
                // effects are masked)
                const unsigned *rr = wrec + r * kRecDw;
                const uint4 r0 = *reinterpret_cast<const uint4 *>(rr);
                const uint4 r1 = *reinterpret_cast<const uint4 *>(rr + 4);
                const uint4 r2 = *reinterpret_cast<const uint4 *>(rr + 8);
                const float wt[4] = {__builtin_bit_cast(float, r1.x), __builtin_bit_cast(float, r1.y),
                                     __builtin_bit_cast(float, r1.z), __builtin_bit_cast(float, r1.w)};
                const float a = __builtin_bit_cast(float, r2.x), lh = __builtin_bit_cast(float, r2.y), lw = __builtin_bit_cast(float, r2.z);
                const unsigned flags = r2.w;
                float g[CPL];
                RG::widen(graw, g);
                if (on && finite) {
                    const float as = a * scale;
                    float ag[CPL];
#pragma unroll
                    for (int i = 0; i < CPL; ++i) ag[i] = as * g[i];
                    accumulate<CPL>(win, r0.z, r0.w, wt, ag, k, magic);
                }
                if (on && (flags & 0xF0u) && shift == 0) {    // corners nobody else will see: global fp32 atomics (.cuh:125-152)
                    const unsigned pix = r0.y & 0xFFFFFFu;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (flags & (16u << c)) {
                            float *p = far_lev + (static_cast<int64_t>(pix) + ((c & 1) && (r0.y >> 30 & 1u) ? 1 : 0) + ((c >> 1) && (r0.y >> 31) ? W : 0)) * (M * kCH);
#pragma unroll
                            for (int i = 0; i < CPL; ++i) unsafeAtomicAdd(p + i, wt[c] * (a * g[i]));
                        }
                    }
                    if (!OWNER && k == 0) hdr->far = 1u;
                }
                float e[4] = {0.f, 0.f, 0.f, 0.f};
                CornerDots<GT, VT, CPL>::run(graw, g, vraw, e);
                // over the LPS lanes of the sample, every lane takes part
                float d0 = sum_sample<LPS>(e[0]), d1 = sum_sample<LPS>(e[1]), d2 = sum_sample<LPS>(e[2]), d3 = sum_sample<LPS>(e[3]);
                d0 = (flags & 1u) ? d0 : 0.f; d1 = (flags & 2u) ? d1 : 0.f;              // a corner outside the map reads nothing (.cuh:56-78)
                d2 = (flags & 4u) ? d2 : 0.f; d3 = (flags & 8u) ? d3 : 0.f;
                if (on && k == 0) {
                    const float hh = 1.f - lh, hw = 1.f - lw;
                    const unsigned p = (flags >> 8) & 15u;
                    const unsigned o_ = (pair0u + r0.x * static_cast<unsigned>(M)) * static_cast<unsigned>(LP) + static_cast<unsigned>(l * P) + p;   // < 2^28 (plan)
                    grad_attn[o_] = wt[0] * d0 + wt[1] * d1 + wt[2] * d2 + wt[3] * d3;                                     // .cuh:156
                    reinterpret_cast<float2 *>(grad_loc)[o_] = make_float2(static_cast<float>(W) * (a * (hh * (d1 - d0) + lh * (d3 - d2))),   // .cuh:157
                                                                            static_cast<float>(H) * (a * (hw * (d2 - d0) + lw * (d3 - d1))));  // .cuh:158
                }
            };
            // a batch of exactly NG groups (the last one possibly ragged): straight-line code, lanes beyond the list re-read its
            // last record and are masked at the use
            auto own_batch = [&](auto ngc, int i0) __attribute__((always_inline)) {
                constexpr int NG = decltype(ngc)::value;
                typename RG::T rg[NG];
                typename RV::T rv[NG][4];
#pragma unroll
                for (int t = 0; t < NG; ++t) {
                    const int r = min(i0 + SPG * t + j, no - 1);
                    const uint2 h = *reinterpret_cast<const uint2 *>(wrec + r * kRecDw);
                    rg[t] = RG::load(gbase + (pair0u + h.x * static_cast<unsigned>(M)) * static_cast<unsigned>(kCH * Elem<GT>::kBytes));
                    const int dxb = (h.y >> 30) & 1u ? rowb : 0, dyb = (h.y >> 31) ? W * rowb : 0;
                    const char *vb = vlev + (h.y & 0xFFFFFFu) * static_cast<unsigned>(rowb);
                    rv[t][0] = RV::load(vb); rv[t][1] = RV::load(vb + dxb);
                    rv[t][2] = RV::load(vb + dyb); rv[t][3] = RV::load(vb + dyb + dxb);
                }
#pragma unroll
                for (int t = 0; t < NG; ++t) {
                    const int r = i0 + SPG * t + j;
                    run_own(r < no, min(r, no - 1), rg[t], rv[t]);
                }
            };
            for (int i0 = 0; i0 < no; i0 += SPG * GMAX) {
                const int ng = min(GMAX, (no - i0 + SPG - 1) / SPG);                      // wave-uniform
                if constexpr (GMAX == 1) {
                    own_batch(std::integral_constant<int, 1>(), i0);
                } else {
                    if (ng == 1) own_batch(std::integral_constant<int, 1>(), i0);
                    else if (GMAX == 2 || ng == 2) own_batch(std::integral_constant<int, 2>(), i0);
                    else if constexpr (GMAX >= 4) {
                        if (ng == 3) own_batch(std::integral_constant<int, 3>(), i0);
                        else own_batch(std::integral_constant<int, 4>(), i0);
                    }
                }
            }
            // ---- b2. neighbours' samples that reach into this core: accumulate only ---------------------------------------
            auto halo_batch = [&](auto ngc, int i0) __attribute__((always_inline)) {
                constexpr int NG = decltype(ngc)::value;
                typename RG::T rg[NG];
                const int h0 = 64 - nh;
#pragma unroll
                for (int t = 0; t < NG; ++t) {
                    const int r = h0 + min(i0 + SPG * t + j, nh - 1);
                    rg[t] = RG::load(gbase + (pair0u + wrec[r * kRecDw] * static_cast<unsigned>(M)) * static_cast<unsigned>(kCH * Elem<GT>::kBytes));
                }
#pragma unroll
                for (int t = 0; t < NG; ++t) {
                    if (i0 + SPG * t + j < nh) {
                        const unsigned *rr = wrec + (h0 + i0 + SPG * t + j) * kRecDw;
                        const uint4 r0 = *reinterpret_cast<const uint4 *>(rr);
                        const uint4 r1 = *reinterpret_cast<const uint4 *>(rr + 4);
                        const float wt[4] = {__builtin_bit_cast(float, r1.x), __builtin_bit_cast(float, r1.y),
                                             __builtin_bit_cast(float, r1.z), __builtin_bit_cast(float, r1.w)};
                        const float as = __builtin_bit_cast(float, rr[8]) * scale;
                        float ag[CPL];
                        RG::widen(rg[t], ag);
#pragma unroll
                        for (int i = 0; i < CPL; ++i) ag[i] *= as;
                        accumulate<CPL>(win, r0.z, r0.w, wt, ag, k, magic);
                    }
                }
            };
            constexpr int HG = LPS == 2 ? 2 : 4;              // (a neighbour list is at most 64 samples = 2 groups of 32)
            for (int i0 = 0; i0 < nh; i0 += HG * SPG) {
                const int ng = min(HG, (nh - i0 + SPG - 1) / SPG);
                if (ng == 1) halo_batch(std::integral_constant<int, 1>(), i0);
                else if (HG == 2 || ng == 2) halo_batch(std::integral_constant<int, 2>(), i0);
                else if constexpr (HG >= 4) {
                    if (ng == 3) halo_batch(std::integral_constant<int, 3>(), i0);
                    else halo_batch(std::integral_constant<int, 4>(), i0);
                }
            }
            wave_sync();
            base = next_base;
        }
        __syncthreads();

        // ---- c. read-out: strip the n * (magic bits) the atomics added along, store.  Did any cell draw more contributions than a
        //         32-bit field holds at this scale?  The maximum count is taken along; the (rare) repeat overwrites what this pass stored.
        //         (Owner scheme: the read-out ADDS into grad_value, so its counts are checked first.)
        if (OWNER) {
            unsigned mc = 0u;
            for (int i = threadIdx.x; i < w.ncell; i += THREADS) mc = max(mc, cnt[i]);
            for (int o = 32; o > 0; o >>= 1) mc = max(mc, __shfl_xor(mc, o));
            if (lane == 0) atomicMax(blk, mc);
            __syncthreads();
            const unsigned maxcnt0 = blk[0];
            __syncthreads();
            if (finite && maxcnt0 >= (512u << shift) && shift < 12) continue;
        }
        {
            const unsigned long long cbits = static_cast<unsigned long long>(__builtin_bit_cast(unsigned, magic)) * 0x100000001ull;
            const float inv = finite ? ldexpf(1.0f, -(22 - shift - e)) : 0.f;
            float *dst1 = scratch + (static_cast<int64_t>(b) * M + m) * pl.scr_per_bm + pl.scr0[l] + static_cast<int64_t>(w.slot) * w.ncell * kCH;
            float *dst0 = grad_value + ((static_cast<int64_t>(b) * pl.S + pl.start[l]) * M + m) * kCH;
            unsigned mc = 0u;
            for (int i = threadIdx.x; i < w.ncell * kCellU64; i += THREADS) {
                const int cell = i / kCellU64, pr = i % kCellU64;
                const unsigned n = cnt[cell];
                mc = max(mc, n);
                const unsigned long long s = win[cell * kCellStride + pr] - static_cast<unsigned long long>(n) * cbits;
                const int lo = static_cast<int>(static_cast<unsigned>(s));
                const int hi = static_cast<int>((static_cast<long long>(s) - static_cast<long long>(lo)) >> 32);
                const float2 out = make_float2(static_cast<float>(lo) * inv, static_cast<float>(hi) * inv);
                if (OWNER) {
                    // this block's share of the cell (its own samples only): added to the shares of the up to eight neighbours whose
                    // halo covers it, and of the other query chunks (mode 1) -- fp32 atomics into the zero-filled grad_value, as the
                    // reference accumulates (.cuh:125-152), but one per cell, channel and block instead of one per sample
                    const int yy = w.cy0 + cell / w.tstride, xx = w.cx0 + cell % w.tstride;
                    if (n != 0u && yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        float *dst = dst0 + static_cast<int64_t>(yy * W + xx) * (M * kCH) + 2 * pr;
                        unsafeAtomicAdd(dst, out.x);
                        unsafeAtomicAdd(dst + 1, out.y);
                    }
                } else if (w.mode == 1) {
                    *reinterpret_cast<float2 *>(dst1 + static_cast<int64_t>(cell) * kCH + 2 * pr) = out;
                } else {
                    const int wxx = cell % w.tstride;
                    const int yy = w.cy0 + cell / w.tstride, xx = w.cx0 + wxx;
                    if (yy < H && xx < W && wxx < pl.TW[l])   // cores partition the level: exclusive owner, plain store (not the padding column)
                        *reinterpret_cast<float2 *>(dst0 + static_cast<int64_t>(yy * W + xx) * (M * kCH) + 2 * pr) = out;
                }
            }
            if (OWNER) break;
            for (int o = 32; o > 0; o >>= 1) mc = max(mc, __shfl_xor(mc, o));
            // (one vote word per pass parity: the word of pass `shift` is zeroed again at the start of pass shift + 2, behind a barrier
            // of pass shift + 1 -- a wave still reading it cannot meet the reset.  No __syncthreads_or: its library form takes
            // static LDS on top of the 160 KB this kernel asks for, and the launch fails.)
            if (lane == 0) atomicMax(blk + 2 + (shift & 1), mc);
            __syncthreads();
            const unsigned maxcnt = blk[2 + (shift & 1)];
            if (!(finite && maxcnt >= (512u << shift) && shift < 12)) {
                break;
            }
        }
    }
}

// ---- 3. finalize: sum the query chunks of the small levels; fold the `far` buffer in if it was used ------------------
// Grid-stride over the rows that need anything: the chunked levels' rows always (a few percent of the pyramid), every row only
// when some block used the `far` buffer (a device flag: not known at launch time).
__device__ __forceinline__ void finalize_row(const FusedPlan &pl, const float *__restrict__ scratch, float *__restrict__ far,
                                             float *__restrict__ grad_value, int l, int b, int pix, int m, int c4, bool use_far)
{
    const bool chunks = pl.mode[l] == 1;
    const int64_t row = (static_cast<int64_t>(b) * pl.S + pix) * pl.M + m;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (chunks) {
        const int ncell = pl.H[l] * pl.W[l];
        const float *base = scratch + (static_cast<int64_t>(b) * pl.M + m) * pl.scr_per_bm + pl.scr0[l] +
                            static_cast<int64_t>(pix - pl.start[l]) * kCH + c4;
        const int64_t cs = static_cast<int64_t>(ncell) * kCH;
        const int nch = pl.nchunk[l];
        // fixed order (deterministic); up to 12 requests in flight -- the default plan's 12 chunks in ONE round trip (the pass moves
        // 59 MB with ~300 k threads: what it waits for is round trips, with four requests at a time there were three)
        constexpr int kFlight = 12;
        for (int t0 = 0; t0 < nch; t0 += kFlight) {
            float4 v[kFlight];
#pragma unroll
            for (int u = 0; u < kFlight; ++u)
                if (t0 + u < nch) v[u] = *reinterpret_cast<const float4 *>(base + (t0 + u) * cs);
#pragma unroll
            for (int u = 0; u < kFlight; ++u)
                if (t0 + u < nch) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    } else {
        acc = *reinterpret_cast<const float4 *>(grad_value + row * kCH + c4);
    }
    if (use_far) {
        float4 *f = reinterpret_cast<float4 *>(far + row * kCH + c4);
        const float4 v = *f;
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        *f = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    *reinterpret_cast<float4 *>(grad_value + row * kCH + c4) = acc;
}

__global__ __launch_bounds__(256)
void msda_finalize_kernel(const FusedPlan pl, const float *__restrict__ scratch, float *__restrict__ far,
                          float *__restrict__ grad_value, Header *__restrict__ hdr)
{
    // 8 lanes x float4 per (b, pixel, m) row of 32 channels
    const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    if (gid == 0) {                                          // from here on the `far` buffer is all zero between calls
        hdr->cookie = kCookie;
        hdr->far_elems = static_cast<unsigned long long>(pl.B) * pl.S * pl.M * kCH;
    }
    const bool use_far = hdr->far != 0u;
    if (use_far) {                                           // every row (the chunked levels' included)
        const int64_t n = static_cast<int64_t>(pl.B) * pl.S * pl.M * 8;
        const bool narrow = n < (1LL << 34);                 // (uniform) row numbers fit 32 bits: no 64-bit divisions (~150 instructions each)
        for (int64_t i = gid; i < n; i += stride) {
            const int64_t row = i >> 3;
            int m, pix, b;
            if (narrow) {
                const unsigned r = static_cast<unsigned>(row), bp = r / static_cast<unsigned>(pl.M);
                m = static_cast<int>(r - bp * static_cast<unsigned>(pl.M));
                b = static_cast<int>(bp / static_cast<unsigned>(pl.S));
                pix = static_cast<int>(bp - static_cast<unsigned>(b) * static_cast<unsigned>(pl.S));
            } else {
                m = static_cast<int>(row % pl.M);
                const int64_t bp = row / pl.M;
                pix = static_cast<int>(bp % pl.S); b = static_cast<int>(bp / pl.S);
            }
            int l = 0;
            while (l + 1 < pl.L && pix >= pl.start[l + 1]) ++l;
            finalize_row(pl, scratch, far, grad_value, l, b, pix, m, static_cast<int>(i & 7) * 4, true);
        }
        return;
    }
    // the chunked levels' rows, all levels in ONE index space (level after level; a thread rarely sees more than one row): rows in
    // the order of the partial windows, [b][m][cell] -- the 8 rows of a wave are 8 consecutive cells = 1 KB contiguous in every
    // chunk (with the head fastest they were 8 windows apart: 128-byte reads, 52 us for 59 MB)
    int64_t cum[kMaxLevels + 1];
    cum[0] = 0;
#pragma unroll
    for (int l = 0; l < kMaxLevels; ++l)
        cum[l + 1] = cum[l] + ((l < pl.L && pl.mode[l] == 1) ? static_cast<int64_t>(pl.B) * pl.H[l] * pl.W[l] * pl.M * 8 : 0);
    const bool narrow = cum[kMaxLevels] < (1LL << 34);
    for (int64_t i = gid; i < cum[kMaxLevels]; i += stride) {
        int l = 0;
#pragma unroll
        for (int t = 1; t < kMaxLevels; ++t) l += i >= cum[t] ? 1 : 0;
        const int hw = pl.H[l] * pl.W[l];
        const int64_t row = (i - cum[l]) >> 3;
        int cell, b, m;
        if (narrow) {
            const unsigned r = static_cast<unsigned>(row), bm = r / static_cast<unsigned>(hw);
            cell = static_cast<int>(r - bm * static_cast<unsigned>(hw));
            b = static_cast<int>(bm / static_cast<unsigned>(pl.M));
            m = static_cast<int>(bm - static_cast<unsigned>(b) * static_cast<unsigned>(pl.M));
        } else {
            cell = static_cast<int>(row % hw);
            const int64_t bm = row / hw;
            b = static_cast<int>(bm / pl.M); m = static_cast<int>(bm % pl.M);
        }
        finalize_row(pl, scratch, far, grad_value, l, b, pl.start[l] + cell, m, static_cast<int>(i & 7) * 4, false);
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------
int env_int(const char *key, int dflt) { return tune_int(key, dflt); }      // MDETR_TUNE (mdetr_tune.h): tests force a tile / wave geometry

bool build_plan(FusedPlan &pl, const int64_t *shapes_h, const int64_t *start_h, int B, int S, int M, int L, int Lq, int P, int elem_dtype = 2)
{
    if (L < 1 || L > kMaxLevels || P < 1 || P > 8 || B < 1 || M < 1 || Lq < 1) return false;
    memset(&pl, 0, sizeof(pl));
    pl.B = B; pl.S = S; pl.M = M; pl.L = L; pl.P = P; pl.Lq = Lq;
    int64_t total = 0;
    for (int l = 0; l < L; ++l) {
        pl.H[l] = static_cast<int>(shapes_h[2 * l]);
        pl.W[l] = static_cast<int>(shapes_h[2 * l + 1]);
        pl.start[l] = static_cast<int>(start_h[l]);
        if (pl.H[l] <= 0 || pl.W[l] <= 0 || pl.start[l] != total) return false;
        if (static_cast<int64_t>(pl.H[l]) * pl.W[l] >= (1 << 24) || pl.H[l] >= (1 << 15) || pl.W[l] >= (1 << 15)) return false;   // packed indices
        total += static_cast<int64_t>(pl.H[l]) * pl.W[l];
    }
    if (total != S) return false;
    // 32-bit index arithmetic in the kernel
    if (static_cast<int64_t>(B) * Lq * M * L * P >= (1 << 28) || static_cast<int64_t>(B) * Lq * M * kCH * 4 >= (1LL << 31) ||
        static_cast<int64_t>(S) * M * kCH * 4 >= (1LL << 31)) return false;
    const bool self = Lq == S;                               // queries = the pyramid's cells, in order
    if (!self && static_cast<int64_t>(Lq) * P > 16384) return false;   // every block scans every query: only for few queries
    // tuning knobs (read per call: a handful of getenv()s against a multi-microsecond launch sequence).  Defaults from the
    // sweeps on MI355X at the encoder shape (profiles/r02f_opbench_*.json, r02o_): 24 x 32 core tiles (bf16, 16 waves: 0.545 ms
    // vs 0.557 at 16 x 40, and 0.72 vs 0.82 for N(0, 4 px) offsets), reach 5 cells (the model's initial offsets reach 4 px on
    // every level), whole-level windows up to 512 cells split into 12 query chunks.
    // (the fp32 form runs 8 waves per workgroup: 24 x 40, 0.88 ms vs 0.95 at 16 x 32)
    const int tile_h = env_int("msda_tile_h", 24), tile_w = env_int("msda_tile_w", elem_dtype == 2 ? 32 : 40);
    const int reach = env_int("msda_reach", 5), chunks_env = env_int("msda_chunks", 12);
    const int whole_max = env_int("msda_whole_level_cells", 512);
    if (tile_h < 1 || tile_w < 1 || tile_h * tile_w > kMaxCells || reach < 0 || chunks_env < 1) return false;
    // (the owner scheme -- a block looks at its own queries only, 16 x 24 core tiles with a 4-cell halo -- lost its A/B in round 3 and
    // is no longer instantiated: the OWNER = true branches of the kernel are compiled out, its switch and launch forms are gone)
    pl.owner = 0;
    const int otile_h = 16, otile_w = 24, oreach = 4;
    if (pl.owner && (otile_h < 1 || otile_w < 1 || oreach < 0 || (otile_h + 2 * oreach) * (otile_w + 2 * oreach) > kMaxCells)) return false;
    pl.small = (reach < (1 << 13) && oreach < (1 << 13)) ? 1 : 0;
    for (int l = 0; l < L; ++l)
        if (pl.H[l] >= (1 << 13) || pl.W[l] >= (1 << 13)) pl.small = 0;
    int blk = 0;
    long long scr = 0;
    pl.max_cells = 0;
    for (int li = 0; li < L; ++li) {
        const int l = L - 1 - li;                            // small, sample-dense levels first: their blocks are the longest
        const int H = pl.H[l], W = pl.W[l];
        int cells;
        if (self && H * W <= whole_max) {
            pl.mode[l] = 1;
            pl.nchunk[l] = chunks_env < Lq ? chunks_env : Lq;
            pl.nblk[l] = pl.nchunk[l];
            cells = H * W;
            pl.scr0[l] = scr;
            if (!pl.owner) scr += static_cast<long long>(pl.nchunk[l]) * cells * kCH;
        } else if (pl.owner) {
            pl.mode[l] = 3;
            const int TH = otile_h < H ? otile_h : H, TW = otile_w < W ? otile_w : W;
            pl.TH[l] = TH; pl.TW[l] = TW; pl.R[l] = oreach;
            pl.nty[l] = (H + TH - 1) / TH;
            pl.ntx[l] = (W + TW - 1) / TW;
            pl.nblk[l] = pl.nty[l] * pl.ntx[l];
            cells = (TH + 2 * oreach) * (TW + 2 * oreach);
            if (static_cast<long long>(Lq) * P >= (1 << 22)) return false;       // float index arithmetic of the query decode
        } else {
            pl.mode[l] = self ? 0 : 2;
            int TH = tile_h < H ? tile_h : H, TW = tile_w < W ? tile_w : W;
            if (!self && H * W <= kMaxCells) { TH = H; TW = W; }
            pl.TH[l] = TH; pl.TW[l] = TW; pl.R[l] = reach;
            pl.nty[l] = (H + TH - 1) / TH;
            pl.ntx[l] = (W + TW - 1) / TW;
            // the owner's "does a neighbour see this query" test looks one tile out: the reach must not exceed a tile
            if (self && ((pl.nty[l] > 1 && reach > TH) || (pl.ntx[l] > 1 && reach > TW))) return false;
            pl.nblk[l] = pl.nty[l] * pl.ntx[l];
            cells = TH * (TW + row_pad(TW));
            if (self) {                                      // centre-cell tables: rows + columns of the candidate rectangle on every query level
                int tab = 0;
                for (int lq = 0; lq < L; ++lq)
                    tab += ((TH + 2 * reach) * pl.H[lq] + H - 1) / H + ((TW + 2 * reach) * pl.W[lq] + W - 1) / W + 4;
                pl.max_tab = tab > pl.max_tab ? tab : pl.max_tab;
                if (static_cast<long long>(Lq) * P >= (1 << 22)) return false;   // float index arithmetic of the candidate decode
            }
        }
        if (cells > kMaxCells || cells >= 0xFFF0) return false;
        pl.max_cells = cells > pl.max_cells ? cells : pl.max_cells;
        pl.bstride[l] = 1;
    }
    // block numbering = dispatch order: the longest blocks first, so that the launch's tail is made of short ones.  Measured per
    // block at the encoder shape (cycles per phase, measured with a timing build in round 3): a 24 x 32 core tile 85 us, a 1/12 query chunk of a whole level
    // 42 us -- with the chunked levels in front (rounds 2 and 3) the last quarter-round of tiles ran on a quarter of the CUs.
    // Chunked levels with the same chunk count stay interleaved chunk by chunk.
    for (int l = 0; l < L; ++l) {                            // finest level first: its tiles are full-sized
        if (pl.mode[l] == 1) continue;
        pl.blk0[l] = blk;
        blk += pl.nblk[l];
    }
    int n1 = 0, first_chunks = 0;
    for (int l = L - 1; l >= 0; --l)
        if (pl.mode[l] == 1) { if (!n1) first_chunks = pl.nchunk[l]; if (pl.nchunk[l] == first_chunks) ++n1; }
    int j = 0;
    for (int l = L - 1; l >= 0; --l) {
        if (pl.mode[l] == 1 && pl.nchunk[l] == first_chunks) { pl.blk0[l] = blk + j; pl.bstride[l] = n1; ++j; }
    }
    blk += n1 * first_chunks;
    for (int l = L - 1; l >= 0; --l) {
        if (pl.mode[l] != 1 || pl.nchunk[l] == first_chunks) continue;
        pl.blk0[l] = blk;
        blk += pl.nblk[l];
    }
    pl.nblocks = blk;
    pl.scr_per_bm = scr;
    return true;
}

int64_t plan_workspace_bytes(const FusedPlan &pl)
{
    if (pl.owner) return kHdrBytes;                          // the header only: no side buffer, no partial windows
    return kHdrBytes + (static_cast<int64_t>(pl.B) * pl.M * pl.scr_per_bm + static_cast<int64_t>(pl.B) * pl.S * pl.M * kCH) * 4;
}

}  // namespace

int64_t msda_fused_workspace_bytes(const int64_t *shapes_h, const int64_t *start_h, int B, int S, int M, int D, int L, int Lq, int P)
{
    FusedPlan pl;
    if (D != kCH || !shapes_h || !start_h || !build_plan(pl, shapes_h, start_h, B, S, M, L, Lq, P)) return 0;
    return plan_workspace_bytes(pl);
}

// value_dtype / grad_dtype: 0 = f32, 2 = bf16.  Writes all three outputs completely (no pre-zeroing needed).  Returns
// hipErrorNotSupported when the geometry does not qualify (the caller falls back to the atomic path).
hipError_t msda_backward_fused_launch(const int64_t *shapes_h, const int64_t *start_h, const void *value, const float *loc,
                                      const float *attn, const void *grad_out, float *grad_value, float *grad_loc, float *grad_attn,
                                      void *workspace, int64_t workspace_bytes, int B, int S, int M, int D, int L, int Lq, int P,
                                      int elem_dtype, hipStream_t st)
{
    FusedPlan pl;
    if (D != kCH || !shapes_h || !start_h || !build_plan(pl, shapes_h, start_h, B, S, M, L, Lq, P, elem_dtype)) return hipErrorNotSupported;
    if (!workspace || workspace_bytes < plan_workspace_bytes(pl)) return hipErrorNotSupported;
    if (elem_dtype != 0 && elem_dtype != 2) return hipErrorNotSupported;
    Header *hdr = static_cast<Header *>(workspace);
    const int64_t nfar = static_cast<int64_t>(B) * S * M * kCH;
    // candidate scheme: [kHdrBytes, kHdrBytes + 4 nfar) = the side buffer, zero between calls, then the partial windows; owner scheme: neither
    float *far = pl.owner ? grad_value : reinterpret_cast<float *>(static_cast<unsigned char *>(workspace) + kHdrBytes);
    float *scratch = pl.owner ? nullptr : far + nfar;
    hipError_t err;
    const int64_t n_go = static_cast<int64_t>(B) * Lq * M * D, n_at = static_cast<int64_t>(B) * Lq * M * L * P;
    if ((L * P) % 4 != 0) return hipErrorNotSupported;       // the pre-pass reads attn in 16-byte pieces
    // grid-stride over 16-byte pieces, four per lane and trip: no more workgroups than that gives work to (small calls)
    const int64_t pre_items = (n_go + n_at) / 4 + (pl.owner ? nfar / 4 : nfar / 4 / 64);
    const unsigned pre_blocks = static_cast<unsigned>(pre_items / (256 * 4) < kMaxPre ? (pre_items / (256 * 4) > 0 ? pre_items / (256 * 4) : 1) : kMaxPre);
    pl.npre = static_cast<int>(pre_blocks);
    profile_begin(7, Lq, st);
    if (elem_dtype == 2)
        hipLaunchKernelGGL(msda_absmax_kernel<__hip_bfloat16>, dim3(pre_blocks), dim3(256), 0, st, static_cast<const __hip_bfloat16 *>(grad_out), n_go, attn, n_at, hdr, far, nfar, pl.owner);
    else
        hipLaunchKernelGGL(msda_absmax_kernel<float>, dim3(pre_blocks), dim3(256), 0, st, static_cast<const float *>(grad_out), n_go, attn, n_at, hdr, far, nfar, pl.owner);
    profile_end(st);
    // 512 threads (8 waves) or 1024 (16 waves: twice the record buffers, more loads in flight per CU)
    int threads = env_int("msda_threads", 1024);          // (bf16: 0.74 ms at 16 waves vs 0.96 at 8, same tile)
    // (16 waves leave 128 VGPRs: the fp32 form's load batches do not fit; 768 = 12 waves with 170 VGPRs, bf16 only)
    threads = (threads >= 1024 && elem_dtype == 2) ? 1024 : ((threads == 768 && elem_dtype == 2) ? 768 : 512);
    auto lds_bytes = [&](int thr) {
        return lds_win_bytes(pl.max_cells) + lds_cnt_bytes(pl.max_cells) + static_cast<size_t>(thr / 64) * 64 * kRecDw * 4 +
               static_cast<size_t>(pl.max_tab) * 4 + 128 + 16 + 128;
    };
    size_t lds = lds_bytes(threads);
    if (lds > 160 * 1024 && threads > 512) {
        threads = 512;
        lds = lds_bytes(threads);
    }
    if (lds > 160 * 1024) return hipErrorNotSupported;
    int dev = 0;
    if ((err = hipGetDevice(&dev)) != hipSuccess) return err;
    // lanes per sample: 4 (bf16 at 16 waves: 8 channels per lane, two groups of 16 own samples in flight) or 8 (4 channels per
    // lane, four groups of 8)
    int lps = env_int("msda_lps", 4);
    // (2: half a row per lane, one group of 32 own samples in flight -- the per-sample instructions halve again; 16 waves, bf16)
    lps = ((lps == 4 || lps == 2) && elem_dtype == 2 && threads > 512) ? lps : 8;
    if (lps == 2 && threads != 1024) lps = 4;
    const int groups = env_int("msda_groups", 2);      // (12-wave form: 2 or 4 groups of 16 own samples in flight)
    static bool attr_set[10][64] = {};                        // per kernel instance and device
    int which = elem_dtype != 2 ? 0 : (threads == 512 ? 1 : (threads == 768 ? (groups >= 4 ? 5 : 4) : (lps == 8 ? 2 : (lps == 2 ? 9 : 3))));
    typedef __hip_bfloat16 bf;
    const void *kern = which == 0 ? reinterpret_cast<const void *>(msda_bwd_fused<float, float, 512, 4, 8>)
                     : which == 1 ? reinterpret_cast<const void *>(msda_bwd_fused<bf, bf, 512, 4, 8>)
                     : which == 2 ? reinterpret_cast<const void *>(msda_bwd_fused<bf, bf, 1024, 4, 8>)
                     : which == 3 ? reinterpret_cast<const void *>(msda_bwd_fused<bf, bf, 1024, 2, 4>)
                     : which == 4 ? reinterpret_cast<const void *>(msda_bwd_fused<bf, bf, 768, 2, 4>)
                     : which == 9 ? reinterpret_cast<const void *>(msda_bwd_fused<bf, bf, 1024, 1, 2>)
                                  : reinterpret_cast<const void *>(msda_bwd_fused<bf, bf, 768, 4, 4>);
    const int slot_ = which;
    if (dev < 0 || dev >= 64 || !attr_set[slot_][dev]) {
        if ((err = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return err;
        if (dev >= 0 && dev < 64) attr_set[slot_][dev] = true;
    }
    const unsigned nblocks = static_cast<unsigned>(B) * M * pl.nblocks;
    profile_begin(6, Lq, st);
    auto go = [&](auto k, auto vt, auto gt) {
        using VT = decltype(vt); using GT = decltype(gt);
        hipLaunchKernelGGL(k, dim3(nblocks), dim3(threads), lds, st, pl, static_cast<const VT *>(value), loc, attn,
                           static_cast<const GT *>(grad_out), grad_value, grad_loc, grad_attn, hdr, scratch, far);
    };
    if (which == 0) go(msda_bwd_fused<float, float, 512, 4, 8>, float(), float());
    else if (which == 1) go(msda_bwd_fused<bf, bf, 512, 4, 8>, bf(), bf());
    else if (which == 2) go(msda_bwd_fused<bf, bf, 1024, 4, 8>, bf(), bf());
    else if (which == 3) go(msda_bwd_fused<bf, bf, 1024, 2, 4>, bf(), bf());
    else if (which == 4) go(msda_bwd_fused<bf, bf, 768, 2, 4>, bf(), bf());
    else if (which == 9) go(msda_bwd_fused<bf, bf, 1024, 1, 2>, bf(), bf());
    else go(msda_bwd_fused<bf, bf, 768, 4, 4>, bf(), bf());
    profile_end(st);
    if (pl.owner) return hipGetLastError();                  // every block added its share into grad_value: nothing to finalize
    const int64_t nrows = static_cast<int64_t>(B) * S * M;
    profile_begin(8, Lq, st);
    const int64_t fin_blocks = (nrows * 8 + 255) / 256;      // grid-stride: at most 2 048 workgroups
    hipLaunchKernelGGL(msda_finalize_kernel, dim3(static_cast<unsigned>(fin_blocks < 2048 ? fin_blocks : 2048)), dim3(256), 0, st, pl, scratch, far, grad_value, hdr);
    profile_end(st);
    return hipGetLastError();
}

}  // namespace mdetr
