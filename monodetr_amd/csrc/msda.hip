// monodetr_amd/csrc/msda.hip -- multi-scale deformable attention for MI355X (gfx950, wave64).
//
// What it computes (semantics fixed by the reference, /root/reference/lib/models/monodetr/ops):
//   forward   src/cuda/ms_deform_im2col_cuda.cuh:237-299 + :33-84
//   backward  src/cuda/ms_deform_im2col_cuda.cuh:301-403 + :87-159 (and the _gm variant :845-920)
// How it computes it is new.  The reference runs one thread per output scalar (1024-thread blocks
// forward, D-thread = half-wave blocks backward with serial shared-memory sums).  Here:
//
//   fast path (f32, D == 32, the shipped geometry M*D = 256):
//     - a "pair" is one (b, q, m); its 32 channels are one 128-byte row of `value` per pixel.
//       8 lanes x float4 cover that row, so a wave64 handles 8 consecutive pairs and every corner
//       gather is ONE dwordx4 instruction fetching eight full 128-B lines (1 KiB / instruction).
//     - loc / attn of the 8 pairs are contiguous in memory: one coalesced dwordx4 (+ one
//       dwordx4 on half the lanes) brings them in, a wave-private LDS slab (padded so the eight
//       broadcast reads hit distinct banks) hands them to the 8 lanes of each pair.  The reference
//       re-reads them from global memory in every one of the D channel threads.
//     - blockIdx -> (image, query chunk) is chosen so all blocks that the dispatcher places on one
//       XCD (block b -> XCD b % 8) walk the queries of the same image in order: each XCD's private
//       4 MiB L2 then holds a sliding band of one image's value rows instead of all B images.
//     - backward: grad_loc / grad_attn are reduced over the 8 lanes of a pair with three DPP adds
//       (quad_perm, quad_perm, row_half_mirror) -- no barriers, no serial sums -- staged in the LDS
//       slab and written back coalesced; grad_value goes out as hardware fp32 atomics
//       (global_atomic_add_f32, executed in L2).
//   generic path (any D, f32 / f64): one thread per (b, q, m, c), all reductions by atomics, like
//     the reference's `_gm` kernel.  Used for the reference's gradcheck shapes (D = 30 .. 3096).
//
// Pixel coordinate: `loc * size - 0.5` with the product ROUNDED before the subtraction -- the
// reference's 0.5 is a double literal (.cuh:285-286), so no FMA there -- see pix_coord().
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mdetr_wave.h>

#include "msda.h"
#include "mdetr_tune.h"

namespace mdetr {
namespace {

// ------------------------------------------------------------------------------------------------
// shared arithmetic
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T pix_coord(T loc, int size)
{
#pragma clang fp contract(off)
    const T prod = loc * static_cast<T>(size);
    return prod - static_cast<T>(0.5);
}

template <typename T>
struct Foot {
    int h_low, w_low;
    T lh, lw, hh, hw;
    bool ok1, ok2, ok3, ok4;   // (low,low) (low,high) (high,low) (high,high), false outside window
    bool inwin;
};

template <typename T>
__device__ __forceinline__ Foot<T> footprint(T h_im, T w_im, int H, int W)
{
    Foot<T> f;
    f.inwin = h_im > T(-1) && w_im > T(-1) && h_im < static_cast<T>(H) && w_im < static_cast<T>(W);   // .cuh:288
    // keep the float->int conversion defined for wild locations: outside the window the values are unused
    const T hs = f.inwin ? h_im : T(0), ws = f.inwin ? w_im : T(0);
    const T hf = floor(hs), wf = floor(ws);
    f.h_low = static_cast<int>(hf);
    f.w_low = static_cast<int>(wf);
    f.lh = hs - hf;
    f.lw = ws - wf;
    f.hh = T(1) - f.lh;
    f.hw = T(1) - f.lw;
    const bool hl = f.h_low >= 0, wl = f.w_low >= 0, hh_ = f.h_low + 1 <= H - 1, wh_ = f.w_low + 1 <= W - 1;
    f.ok1 = f.inwin && hl && wl;
    f.ok2 = f.inwin && hl && wh_;
    f.ok3 = f.inwin && hh_ && wl;
    f.ok4 = f.inwin && hh_ && wh_;
    return f;
}

// component-wise select (a whole-float4 ?: is lowered through scratch memory by hipcc)
__device__ __forceinline__ float4 keep4(bool ok, const float4 &v)
{
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// sum over the 8 lanes of a pair group; result in all 8 lanes
__device__ __forceinline__ float sum8(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    return v;
}

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS operations of one wave execute in order; this only stops the compiler from moving
    // the slab reads above the slab writes (and the next round's writes above this round's reads).
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

constexpr int kWaves = 4;          // 256-thread workgroups
constexpr int kPairsPerWave = 8;   // 8 lanes x float4 = one 128-B (D = 32) row per pair
constexpr int kPad = 4;            // floats of padding per pair in the slab (keeps 16-B alignment)

__host__ __device__ constexpr int slab_floats(int LP) { return kPairsPerWave * (2 * LP + kPad) + kPairsPerWave * (LP + kPad); }

// Stage loc / attn (or nothing but geometry) of `nv` pairs starting at global pair `g0` into the
// wave's slab.  LP % 4 == 0, so a float4 never straddles two pairs.
template <int TLP>
__device__ __forceinline__ void stage_pairs(const float *__restrict__ loc, const float *__restrict__ attn,
                                            float *s_loc, float *s_att, int64_t g0, int nv, int LP_, int lane)
{
    const int LP = TLP ? TLP : LP_;
    const float4 *gl = reinterpret_cast<const float4 *>(loc + g0 * 2 * LP);
    const float4 *ga = reinterpret_cast<const float4 *>(attn + g0 * LP);
    const int nl4 = nv * 2 * LP / 4, na4 = nv * LP / 4;
    for (int i = lane; i < nl4; i += 64) {
        const float4 v = gl[i];
        const int f = 4 * i, j = f / (2 * LP), r = f - j * 2 * LP;
        *reinterpret_cast<float4 *>(s_loc + j * (2 * LP + kPad) + r) = v;
    }
    for (int i = lane; i < na4; i += 64) {
        const float4 v = ga[i];
        const int f = 4 * i, j = f / LP, r = f - j * LP;
        *reinterpret_cast<float4 *>(s_att + j * (LP + kPad) + r) = v;
    }
}

struct LevelGeom { int H, W; int start; };

__device__ __forceinline__ LevelGeom level_geom(const int64_t *__restrict__ shapes, const int64_t *__restrict__ lstart, int l)
{
    LevelGeom g;
    g.H = static_cast<int>(shapes[2 * l]);          // wave-uniform -> scalar loads
    g.W = static_cast<int>(shapes[2 * l + 1]);
    g.start = static_cast<int>(lstart[l]);
    return g;
}

// acc += bilinear(v1..v4) * attn for one sample, 4 channels per lane (.cuh:80-82, :290)
__device__ __forceinline__ void accumulate(float4 &acc, const Foot<float> &f, float a,
                                           float4 v1, float4 v2, float4 v3, float4 v4)
{
    v1 = keep4(f.ok1, v1); v2 = keep4(f.ok2, v2); v3 = keep4(f.ok3, v3); v4 = keep4(f.ok4, v4);
    const float w1 = f.hh * f.hw, w2 = f.hh * f.lw, w3 = f.lh * f.hw, w4 = f.lh * f.lw;
    const float wa = f.inwin ? a : 0.f;
    acc.x += (w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x) * wa;
    acc.y += (w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y) * wa;
    acc.z += (w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z) * wa;
    acc.w += (w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w) * wa;
}

// ------------------------------------------------------------------------------------------------
// fast path, forward
// ------------------------------------------------------------------------------------------------
template <int TL, int TP>
__global__ __launch_bounds__(kWaves * 64)
void msda_fwd_d32(const float *__restrict__ value, const int64_t *__restrict__ shapes,
                  const int64_t *__restrict__ lstart, const float *__restrict__ loc,
                  const float *__restrict__ attn, float *__restrict__ out,
                  int B, int S, int M, int L_, int P_, int npairs, int iters)
{
    MDETR_DYNAMIC_LDS(float, smem);
    const int L = TL ? TL : L_, P = TP ? TP : P_, LP = L * P;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x % B, chunk = blockIdx.x / B;   // same image on the same XCD (bid % 8)
    float *s_loc = smem + wave * slab_floats(LP);
    float *s_att = s_loc + kPairsPerWave * (2 * LP + kPad);
    const int j = lane >> 3, k = lane & 7;
    const int64_t img = static_cast<int64_t>(b) * S * M * 32;
    const int row = M * 32;                                  // floats per pixel

    for (int it = 0; it < iters; ++it) {
        const int p0 = (chunk * iters + it) * (kWaves * kPairsPerWave) + wave * kPairsPerWave;
        if (p0 >= npairs) break;                             // wave-uniform
        const int nv = min(kPairsPerWave, npairs - p0);
        const int64_t g0 = static_cast<int64_t>(b) * npairs + p0;
        wave_lds_fence();
        stage_pairs<TL * TP>(loc, attn, s_loc, s_att, g0, nv, LP, lane);
        wave_lds_fence();
        if (j < nv) {
            const int m = (p0 + j) % M;
            const float *vb = value + img + m * 32 + k * 4;
            const float *my_loc = s_loc + j * (2 * LP + kPad);
            const float *my_att = s_att + j * (LP + kPad);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (TP != 0) {
                // compile-time P: issue all 4*P corner gathers of a level before consuming any
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const LevelGeom g = level_geom(shapes, lstart, l);
                    const float *vl = vb + static_cast<int64_t>(g.start) * row;
                    Foot<float> f[TP];
                    float a[TP];
                    float4 v[TP][4];
#pragma unroll
                    for (int p = 0; p < TP; ++p) {
                        const int s = l * TP + p;
                        const float2 xy = *reinterpret_cast<const float2 *>(my_loc + 2 * s);
                        a[p] = my_att[s];
                        f[p] = footprint(pix_coord(xy.y, g.H), pix_coord(xy.x, g.W), g.H, g.W);
                        const int y0 = clampi(f[p].h_low, 0, g.H - 1), y1 = clampi(f[p].h_low + 1, 0, g.H - 1);
                        const int x0 = clampi(f[p].w_low, 0, g.W - 1), x1 = clampi(f[p].w_low + 1, 0, g.W - 1);
                        v[p][0] = *reinterpret_cast<const float4 *>(vl + (y0 * g.W + x0) * row);
                        v[p][1] = *reinterpret_cast<const float4 *>(vl + (y0 * g.W + x1) * row);
                        v[p][2] = *reinterpret_cast<const float4 *>(vl + (y1 * g.W + x0) * row);
                        v[p][3] = *reinterpret_cast<const float4 *>(vl + (y1 * g.W + x1) * row);
                    }
#pragma unroll
                    for (int p = 0; p < TP; ++p) accumulate(acc, f[p], a[p], v[p][0], v[p][1], v[p][2], v[p][3]);
                }
            } else {
                for (int l = 0; l < L; ++l) {
                    const LevelGeom g = level_geom(shapes, lstart, l);
                    const float *vl = vb + static_cast<int64_t>(g.start) * row;
                    for (int p = 0; p < P; ++p) {
                        const int s = l * P + p;
                        const float2 xy = *reinterpret_cast<const float2 *>(my_loc + 2 * s);
                        const float a = my_att[s];
                        const Foot<float> f = footprint(pix_coord(xy.y, g.H), pix_coord(xy.x, g.W), g.H, g.W);
                        const int y0 = clampi(f.h_low, 0, g.H - 1), y1 = clampi(f.h_low + 1, 0, g.H - 1);
                        const int x0 = clampi(f.w_low, 0, g.W - 1), x1 = clampi(f.w_low + 1, 0, g.W - 1);
                        const float4 v1 = *reinterpret_cast<const float4 *>(vl + (y0 * g.W + x0) * row);
                        const float4 v2 = *reinterpret_cast<const float4 *>(vl + (y0 * g.W + x1) * row);
                        const float4 v3 = *reinterpret_cast<const float4 *>(vl + (y1 * g.W + x0) * row);
                        const float4 v4 = *reinterpret_cast<const float4 *>(vl + (y1 * g.W + x1) * row);
                        accumulate(acc, f, a, v1, v2, v3, v4);
                    }
                }
            }
            *reinterpret_cast<float4 *>(out + (g0 + j) * 32 + k * 4) = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fast path, forward, "sample record" form (L * P == 16)
// ------------------------------------------------------------------------------------------------
// In msda_fwd_d32 each of the 8 lanes of a pair recomputes the pair's 16 footprints (coordinates,
// floor, window tests, corner addresses): ~1460 VALU instructions per round, a quarter of them
// selects, ~100 quarter-rate integer multiplies -- PMC (profiles/r01h_pmc_msda.json): the kernel
// issues VALU 38 % of its wave cycles and parks on memory only 38 %.  Here every (pair, sample)
// footprint is computed ONCE, by one lane: a round has 8 x 16 = 128 samples = 2 per lane, whose
// (x, y) and attention weight are read straight from global memory (consecutive lanes, consecutive
// samples: coalesced), and published as a 32-byte LDS record {4 byte offsets, 4 bilinear weights}
// (+ the attention weight in a side array); the 8 lanes of a pair then read each record as a
// broadcast and only gather and accumulate.  A lane's sample slot -- hence its level geometry -- is
// the same in every round, so H, W and the level start are loaded once per kernel.
//
// Exactness: a corner outside the map gets weight 0 and the address of a corner of the same sample
// that IS inside (so the product is 0 x a value the reference also reads -- identical even for
// non-finite values); a sample with no corner inside (or outside the window, .cuh:288) is flagged in
// bit 0 of its first offset and its contribution is replaced by 0, as the reference skips it.
struct alignas(16) SampleRec { int off[4]; float w[4]; };
constexpr int kRecStride = 17;     // records per pair (16 + 1): the 8 pairs a wave reads in one instruction start 8 banks apart

template <int TL, int TP, typename VT = float, typename OT = float>
__global__ __launch_bounds__(kWaves * 64)
void msda_fwd_rec(const VT *__restrict__ value, const int64_t *__restrict__ shapes,
                  const int64_t *__restrict__ lstart, const float *__restrict__ loc,
                  const float *__restrict__ attn, OT *__restrict__ out,
                  int B, int S, int M, int npairs, int iters)
{
    constexpr int kRowBytes = 32 * Elem<VT>::kBytes;       // one head's 32 channels of a pixel
    constexpr int LP = TL * TP;
    static_assert(LP == 16, "one record slot per lane and half-round");
    __shared__ SampleRec s_rec[kWaves][kPairsPerWave * kRecStride];
    __shared__ float s_att[kWaves][kPairsPerWave * kRecStride];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x % B, chunk = blockIdx.x / B;   // same image on the same XCD (bid % 8)
    SampleRec *wrec = s_rec[wave];
    float *watt = s_att[wave];
    const int j = lane >> 3, k = lane & 7;
    const char *vimg = reinterpret_cast<const char *>(value + static_cast<int64_t>(b) * S * M * 32) + k * (4 * Elem<VT>::kBytes);

    // producer role: sample slot s = lane % 16 of pairs lane / 16 and 4 + lane / 16
    const int ps = lane & 15, pl = ps / TP;
    const int H = static_cast<int>(shapes[2 * pl]), W = static_cast<int>(shapes[2 * pl + 1]);
    const int start = static_cast<int>(lstart[pl]);
    const int dx = M * kRowBytes, dy = W * dx;              // bytes to the next pixel / next row

    for (int it = 0; it < iters; ++it) {
        const int p0 = (chunk * iters + it) * (kWaves * kPairsPerWave) + wave * kPairsPerWave;
        if (p0 >= npairs) break;                             // wave-uniform
        const int nv = min(kPairsPerWave, npairs - p0);
        const int64_t g0 = static_cast<int64_t>(b) * npairs + p0;
        float2 xy[2];
        float at[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int pj = 4 * hf + (lane >> 4);
            const int64_t gs = (g0 + (pj < nv ? pj : 0)) * LP + ps;
            xy[hf] = reinterpret_cast<const float2 *>(loc)[gs];
            at[hf] = attn[gs];
        }
        wave_lds_fence();                                    // the previous round's records have been consumed
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int pj = 4 * hf + (lane >> 4);
            const Foot<float> f = footprint(pix_coord(xy[hf].y, H), pix_coord(xy[hf].x, W), H, W);
            const int m = (p0 + pj) % M;
            const int o1 = ((start + f.h_low * W + f.w_low) * M + m) * kRowBytes;
            const int o2 = o1 + dx, o3 = o1 + dy, o4 = o3 + dx;
            const bool any = f.ok1 || f.ok2 || f.ok3 || f.ok4;
            const int fb = f.ok1 ? o1 : (f.ok2 ? o2 : (f.ok3 ? o3 : o4));
            SampleRec r;
            r.off[0] = any ? (f.ok1 ? o1 : fb) : 1;          // bit 0 = "contributes nothing"
            r.off[1] = any ? (f.ok2 ? o2 : fb) : 0;
            r.off[2] = any ? (f.ok3 ? o3 : fb) : 0;
            r.off[3] = any ? (f.ok4 ? o4 : fb) : 0;
            r.w[0] = f.ok1 ? f.hh * f.hw : 0.f;
            r.w[1] = f.ok2 ? f.hh * f.lw : 0.f;
            r.w[2] = f.ok3 ? f.lh * f.hw : 0.f;
            r.w[3] = f.ok4 ? f.lh * f.lw : 0.f;
            if (pj < nv) {
                SampleRec *dst = wrec + pj * kRecStride + ps;
                *reinterpret_cast<int4 *>(dst->off) = make_int4(r.off[0], r.off[1], r.off[2], r.off[3]);
                *reinterpret_cast<float4 *>(dst->w) = make_float4(r.w[0], r.w[1], r.w[2], r.w[3]);
                watt[pj * kRecStride + ps] = f.inwin ? at[hf] : 0.f;
            }
        }
        wave_lds_fence();
        if (j < nv) {
            const SampleRec *pr = wrec + j * kRecStride;
            const float *pa = watt + j * kRecStride;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int l = 0; l < TL; ++l) {
                int4 o[TP];
                float4 w[TP], v[TP][4];
                float a[TP];
#pragma unroll
                for (int p = 0; p < TP; ++p) {               // all 4*P gathers of a level are issued before any is consumed
                    o[p] = *reinterpret_cast<const int4 *>(pr[l * TP + p].off);
                    w[p] = *reinterpret_cast<const float4 *>(pr[l * TP + p].w);
                    a[p] = pa[l * TP + p];
                    v[p][0] = Elem<VT>::load4(vimg + (o[p].x & ~1));
                    v[p][1] = Elem<VT>::load4(vimg + o[p].y);
                    v[p][2] = Elem<VT>::load4(vimg + o[p].z);
                    v[p][3] = Elem<VT>::load4(vimg + o[p].w);
                }
#pragma unroll
                for (int p = 0; p < TP; ++p) {
                    const bool skip = (o[p].x & 1) != 0;
                    const float cx = w[p].x * v[p][0].x + w[p].y * v[p][1].x + w[p].z * v[p][2].x + w[p].w * v[p][3].x;
                    const float cy = w[p].x * v[p][0].y + w[p].y * v[p][1].y + w[p].z * v[p][2].y + w[p].w * v[p][3].y;
                    const float cz = w[p].x * v[p][0].z + w[p].y * v[p][1].z + w[p].z * v[p][2].z + w[p].w * v[p][3].z;
                    const float cw = w[p].x * v[p][0].w + w[p].y * v[p][1].w + w[p].z * v[p][2].w + w[p].w * v[p][3].w;
                    acc.x += (skip ? 0.f : cx) * a[p];
                    acc.y += (skip ? 0.f : cy) * a[p];
                    acc.z += (skip ? 0.f : cz) * a[p];
                    acc.w += (skip ? 0.f : cw) * a[p];
                }
            }
            Elem<OT>::store4(reinterpret_cast<char *>(out + (g0 + j) * 32 + k * 4), acc);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fast path, backward
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add4(float *p, float s, const float4 &t)
{
    unsafeAtomicAdd(p + 0, s * t.x);
    unsafeAtomicAdd(p + 1, s * t.y);
    unsafeAtomicAdd(p + 2, s * t.z);
    unsafeAtomicAdd(p + 3, s * t.w);
}

// Scatter phase of the backward pass with 32 lanes x 1 channel per pair (2 pairs per wave-round):
// each atomic instruction covers two complete 128-B rows of grad_value.  Reads only the slab
// (loc / attn) and grad_out -- no value loads, so no vmcnt wait sits between the atomics.
template <int TL, int TP, typename GT = float>
__device__ __forceinline__ void scatter_rows32(const int64_t *__restrict__ shapes, const int64_t *__restrict__ lstart,
                                               const GT *__restrict__ grad_out, float *__restrict__ grad_value,
                                               const float *s_loc, const float *s_att, int64_t g0, int p0, int nv,
                                               int64_t img, int M, int L, int P, int lane)
{
    const int LP = L * P, row = M * 32;
    const int c = lane & 31, half = lane >> 5;
#pragma unroll
    for (int r = 0; r < kPairsPerWave / 2; ++r) {
        const int jj = 2 * r + half;
        if (jj < nv) {
            const int m = (p0 + jj) % M;
            const float go = Elem<GT>::load1(grad_out + (g0 + jj) * 32 + c);
            float *gvb = grad_value + img + m * 32 + c;
            const float *my_loc = s_loc + jj * (2 * LP + kPad);
            const float *my_att = s_att + jj * (LP + kPad);
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const LevelGeom g = level_geom(shapes, lstart, l);
                float *gvl = gvb + static_cast<int64_t>(g.start) * row;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const int s = l * P + p;
                    const float2 xy = *reinterpret_cast<const float2 *>(my_loc + 2 * s);
                    const float tgv = go * my_att[s];
                    const Foot<float> f = footprint(pix_coord(xy.y, g.H), pix_coord(xy.x, g.W), g.H, g.W);
                    const int o = (f.h_low * g.W + f.w_low) * row;
                    if (f.ok1) unsafeAtomicAdd(gvl + o, f.hh * f.hw * tgv);
                    if (f.ok2) unsafeAtomicAdd(gvl + o + row, f.hh * f.lw * tgv);
                    if (f.ok3) unsafeAtomicAdd(gvl + o + g.W * row, f.lh * f.hw * tgv);
                    if (f.ok4) unsafeAtomicAdd(gvl + o + g.W * row + row, f.lh * f.lw * tgv);
                }
            }
        }
    }
}

// One sample of the backward pass for the 4 channels of this lane: fp32 atomics into grad_value
// (.cuh:113-152), then d/d(attn), d/d(loc) reduced over the pair's 8 lanes and parked in the slab.
template <bool ATOMICS>
__device__ __forceinline__ void scatter_sample(const Foot<float> &f, float a, const float4 &go,
                                               float4 v1, float4 v2, float4 v3, float4 v4,
                                               float *gvl, const int (&o)[4], const LevelGeom &g,
                                               float *my_loc, float *my_att, int s, int k)
{
    v1 = keep4(f.ok1, v1); v2 = keep4(f.ok2, v2); v3 = keep4(f.ok3, v3); v4 = keep4(f.ok4, v4);
    const float w1 = f.hh * f.hw, w2 = f.hh * f.lw, w3 = f.lh * f.hw, w4 = f.lh * f.lw;
    const float4 tgv = make_float4(go.x * a, go.y * a, go.z * a, go.w * a);      // .cuh:113
    if constexpr (ATOMICS) {
        if (f.ok1) atomic_add4(gvl + o[0], w1, tgv);                            // .cuh:125
        if (f.ok2) atomic_add4(gvl + o[1], w2, tgv);                            // .cuh:134
        if (f.ok3) atomic_add4(gvl + o[2], w3, tgv);                            // .cuh:143
        if (f.ok4) atomic_add4(gvl + o[3], w4, tgv);                            // .cuh:152
    }
    float pa = 0.f, pw = 0.f, ph = 0.f;
#define MDETR_CH(c)                                                                              \
    {                                                                                            \
        const float gh = -f.hw * v1.c - f.lw * v2.c + f.hw * v3.c + f.lw * v4.c; /* .cuh:123-151 */ \
        const float gw = -f.hh * v1.c + f.hh * v2.c - f.lh * v3.c + f.lh * v4.c;                 \
        const float val = w1 * v1.c + w2 * v2.c + w3 * v3.c + w4 * v4.c;                         \
        pa += go.c * val;                                                                        \
        pw += gw * tgv.c;                                                                        \
        ph += gh * tgv.c;                                                                        \
    }
    MDETR_CH(x) MDETR_CH(y) MDETR_CH(z) MDETR_CH(w)
#undef MDETR_CH
    pa = sum8(pa);
    pw = sum8(pw) * static_cast<float>(g.W);                                    // .cuh:157
    ph = sum8(ph) * static_cast<float>(g.H);                                    // .cuh:158
    if (k == 0) {                     // slot s has been read by all 8 lanes (in-order LDS)
        *reinterpret_cast<float2 *>(my_loc + 2 * s) = f.inwin ? make_float2(pw, ph) : make_float2(0.f, 0.f);
        my_att[s] = f.inwin ? pa : 0.f;
    }
}

// VAR 2: grad_loc / grad_attn only (grad_value comes from the tile-privatised path, msda_tiled.hip)
// VAR 0: single phase (gathers, float4-lane atomics and reductions interleaved per sample)
// VAR 1: per round, a scatter phase with full-row atomics (scatter_rows32) followed by the
//        gather/reduce phase -- the vmcnt drain the compiler puts between atomics and the next
//        use of a loaded value then happens once per round instead of once per sample.
template <int TL, int TP, int VAR, typename VT = float, typename GT = float>
__global__ __launch_bounds__(kWaves * 64, (VAR == 2 ? 4 : 1))      // gather-only variant: keep <= 128 VGPRs (4 waves / SIMD)
void msda_bwd_d32(const VT *__restrict__ value, const int64_t *__restrict__ shapes,
                  const int64_t *__restrict__ lstart, const float *__restrict__ loc,
                  const float *__restrict__ attn, const GT *__restrict__ grad_out,
                  float *__restrict__ grad_value, float *__restrict__ grad_loc, float *__restrict__ grad_attn,
                  int B, int S, int M, int L_, int P_, int npairs, int iters, unsigned *__restrict__ absmax2)
{
    MDETR_DYNAMIC_LDS(float, smem);
    const int L = TL ? TL : L_, P = TP ? TP : P_, LP = L * P;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x % B, chunk = blockIdx.x / B;
    float *s_loc = smem + wave * slab_floats(LP);
    float *s_att = s_loc + kPairsPerWave * (2 * LP + kPad);
    const int j = lane >> 3, k = lane & 7;
    const int64_t img = static_cast<int64_t>(b) * S * M * 32;
    const int row = M * 32;
    float amax_g = 0.f, amax_a = 0.f, poison = 0.f;       // VAR 2: max|grad_out|, max|attn| for the tiled scatter's scale

    for (int it = 0; it < iters; ++it) {
        const int p0 = (chunk * iters + it) * (kWaves * kPairsPerWave) + wave * kPairsPerWave;
        if (p0 >= npairs) break;
        const int nv = min(kPairsPerWave, npairs - p0);
        const int64_t g0 = static_cast<int64_t>(b) * npairs + p0;
        wave_lds_fence();
        stage_pairs<TL * TP>(loc, attn, s_loc, s_att, g0, nv, LP, lane);
        wave_lds_fence();
        if constexpr (VAR == 2) {
            if (j < nv) {
                const float *ma = s_att + j * (LP + kPad);
                for (int s = k; s < LP; s += 8) { amax_a = fmaxf(amax_a, fabsf(ma[s])); poison += ma[s] * 0.f; }
            }
        }
        if constexpr (VAR == 1)
            scatter_rows32<TL, TP, GT>(shapes, lstart, grad_out, grad_value, s_loc, s_att, g0, p0, nv, img, M, L, P, lane);
        if (j < nv) {
            const int m = (p0 + j) % M;
            const int64_t voff = img + m * 32 + k * 4;
            float *my_loc = s_loc + j * (2 * LP + kPad);
            float *my_att = s_att + j * (LP + kPad);
            const float4 go = Elem<GT>::load4(reinterpret_cast<const char *>(grad_out + (g0 + j) * 32 + k * 4));
            if constexpr (VAR == 2) {
                amax_g = fmaxf(fmaxf(amax_g, fmaxf(fabsf(go.x), fabsf(go.y))), fmaxf(fabsf(go.z), fabsf(go.w)));
                poison += (go.x + go.y + go.z + go.w) * 0.f;
            }
            if constexpr (TP != 0) {
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const LevelGeom g = level_geom(shapes, lstart, l);
                    const int64_t loff = voff + static_cast<int64_t>(g.start) * row;
                    const VT *vl = value + loff;
                    float *gvl = grad_value + loff;
                    Foot<float> f[TP];
                    float a[TP];
                    int o[TP][4];
                    float4 v[TP][4];
#pragma unroll
                    for (int p = 0; p < TP; ++p) {     // all 4*P gathers of the level in flight together
                        const int s = l * TP + p;
                        const float2 xy = *reinterpret_cast<const float2 *>(my_loc + 2 * s);
                        a[p] = my_att[s];
                        f[p] = footprint(pix_coord(xy.y, g.H), pix_coord(xy.x, g.W), g.H, g.W);
                        const int y0 = clampi(f[p].h_low, 0, g.H - 1), y1 = clampi(f[p].h_low + 1, 0, g.H - 1);
                        const int x0 = clampi(f[p].w_low, 0, g.W - 1), x1 = clampi(f[p].w_low + 1, 0, g.W - 1);
                        o[p][0] = (y0 * g.W + x0) * row; o[p][1] = (y0 * g.W + x1) * row;
                        o[p][2] = (y1 * g.W + x0) * row; o[p][3] = (y1 * g.W + x1) * row;
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[p][c] = Elem<VT>::load4(reinterpret_cast<const char *>(vl + o[p][c]));
                    }
#pragma unroll
                    for (int p = 0; p < TP; ++p)
                        scatter_sample<VAR == 0>(f[p], a[p], go, v[p][0], v[p][1], v[p][2], v[p][3], gvl, o[p], g,
                                       my_loc, my_att, l * TP + p, k);
                }
            } else {
                for (int l = 0; l < L; ++l) {
                    const LevelGeom g = level_geom(shapes, lstart, l);
                    const int64_t loff = voff + static_cast<int64_t>(g.start) * row;
                    const VT *vl = value + loff;
                    float *gvl = grad_value + loff;
                    for (int p = 0; p < P; ++p) {
                        const int s = l * P + p;
                        const float2 xy = *reinterpret_cast<const float2 *>(my_loc + 2 * s);
                        const float a = my_att[s];
                        const Foot<float> f = footprint(pix_coord(xy.y, g.H), pix_coord(xy.x, g.W), g.H, g.W);
                        const int y0 = clampi(f.h_low, 0, g.H - 1), y1 = clampi(f.h_low + 1, 0, g.H - 1);
                        const int x0 = clampi(f.w_low, 0, g.W - 1), x1 = clampi(f.w_low + 1, 0, g.W - 1);
                        const int o[4] = {(y0 * g.W + x0) * row, (y0 * g.W + x1) * row, (y1 * g.W + x0) * row, (y1 * g.W + x1) * row};
                        const float4 v1 = Elem<VT>::load4(reinterpret_cast<const char *>(vl + o[0]));
                        const float4 v2 = Elem<VT>::load4(reinterpret_cast<const char *>(vl + o[1]));
                        const float4 v3 = Elem<VT>::load4(reinterpret_cast<const char *>(vl + o[2]));
                        const float4 v4 = Elem<VT>::load4(reinterpret_cast<const char *>(vl + o[3]));
                        scatter_sample<VAR == 0>(f, a, go, v1, v2, v3, v4, gvl, o, g, my_loc, my_att, s, k);
                    }
                }
            }
        }
        wave_lds_fence();
        // coalesced write-back of the slab (mirror of stage_pairs)
        {
            float4 *gl = reinterpret_cast<float4 *>(grad_loc + g0 * 2 * LP);
            float4 *ga = reinterpret_cast<float4 *>(grad_attn + g0 * LP);
            const int nl4 = nv * 2 * LP / 4, na4 = nv * LP / 4;
            for (int i = lane; i < nl4; i += 64) {
                const int fo = 4 * i, jj = fo / (2 * LP), r = fo - jj * 2 * LP;
                gl[i] = *reinterpret_cast<const float4 *>(s_loc + jj * (2 * LP + kPad) + r);
            }
            for (int i = lane; i < na4; i += 64) {
                const int fo = 4 * i, jj = fo / LP, r = fo - jj * LP;
                ga[i] = *reinterpret_cast<const float4 *>(s_att + jj * (LP + kPad) + r);
            }
        }
    }
    if constexpr (VAR == 2) {
        if (absmax2) {
            if (!(poison == 0.f)) amax_g = __builtin_inff();      // NaN / inf somewhere -> scatter takes the atomic path
            for (int o = 32; o > 0; o >>= 1) { amax_g = fmaxf(amax_g, __shfl_xor(amax_g, o)); amax_a = fmaxf(amax_a, __shfl_xor(amax_a, o)); }
            if (lane == 0) {     // same-address atomics serialise in L2: only waves that raise the maximum issue one
                const unsigned ug = __builtin_bit_cast(unsigned, amax_g), ua = __builtin_bit_cast(unsigned, amax_a);
                if (ug > __hip_atomic_load(absmax2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(absmax2, ug);
                if (ua > __hip_atomic_load(absmax2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(absmax2 + 1, ua);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// generic path (any D, float / double): one thread per (b, q, m, c)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void msda_fwd_generic(const T *__restrict__ value, const int64_t *__restrict__ shapes,
                      const int64_t *__restrict__ lstart, const T *__restrict__ loc,
                      const T *__restrict__ attn, T *__restrict__ out,
                      int S, int M, int D, int L, int Lq, int P, int64_t n)
{
    for (int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; idx < n;
         idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int c = static_cast<int>(idx % D);
        const int64_t samp = idx / D;                     // (b*Lq + q)*M + m
        const int m = static_cast<int>(samp % M);
        const int64_t b = samp / M / Lq;
        const int64_t row = static_cast<int64_t>(M) * D;
        const T *locp = loc + samp * L * P * 2;
        const T *attp = attn + samp * L * P;
        T col = 0;
        for (int l = 0; l < L; ++l) {
            const int H = static_cast<int>(shapes[2 * l]), W = static_cast<int>(shapes[2 * l + 1]);
            const T *vb = value + (b * S + lstart[l]) * row + m * D + c;
            for (int p = 0; p < P; ++p) {
                const T x = locp[(l * P + p) * 2], y = locp[(l * P + p) * 2 + 1], a = attp[l * P + p];
                const Foot<T> f = footprint(pix_coord(y, H), pix_coord(x, W), H, W);
                if (!f.inwin) continue;
                const int64_t o = (static_cast<int64_t>(f.h_low) * W + f.w_low) * row;
                const T v1 = f.ok1 ? vb[o] : T(0);
                const T v2 = f.ok2 ? vb[o + row] : T(0);
                const T v3 = f.ok3 ? vb[o + W * row] : T(0);
                const T v4 = f.ok4 ? vb[o + W * row + row] : T(0);
                col += (f.hh * f.hw * v1 + f.hh * f.lw * v2 + f.lh * f.hw * v3 + f.lh * f.lw * v4) * a;
            }
        }
        out[idx] = col;
    }
}

template <typename T>
__global__ __launch_bounds__(256)
void msda_bwd_generic(const T *__restrict__ value, const int64_t *__restrict__ shapes,
                      const int64_t *__restrict__ lstart, const T *__restrict__ loc,
                      const T *__restrict__ attn, const T *__restrict__ grad_out,
                      T *__restrict__ grad_value, T *__restrict__ grad_loc, T *__restrict__ grad_attn,
                      int S, int M, int D, int L, int Lq, int P, int64_t n)
{
    for (int64_t idx = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; idx < n;
         idx += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int c = static_cast<int>(idx % D);
        const int64_t samp = idx / D;
        const int m = static_cast<int>(samp % M);
        const int64_t b = samp / M / Lq;
        const int64_t row = static_cast<int64_t>(M) * D;
        const T *locp = loc + samp * L * P * 2;
        const T *attp = attn + samp * L * P;
        T *glocp = grad_loc + samp * L * P * 2;
        T *gattp = grad_attn + samp * L * P;
        const T top = grad_out[idx];
        for (int l = 0; l < L; ++l) {
            const int H = static_cast<int>(shapes[2 * l]), W = static_cast<int>(shapes[2 * l + 1]);
            const int64_t base = (b * S + lstart[l]) * row + m * D + c;
            const T *vb = value + base;
            T *gvb = grad_value + base;
            for (int p = 0; p < P; ++p) {
                const T x = locp[(l * P + p) * 2], y = locp[(l * P + p) * 2 + 1], a = attp[l * P + p];
                const Foot<T> f = footprint(pix_coord(y, H), pix_coord(x, W), H, W);
                if (!f.inwin) continue;
                const int64_t o = (static_cast<int64_t>(f.h_low) * W + f.w_low) * row;
                const T tgv = top * a;
                T gh = 0, gw = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                if (f.ok1) { v1 = vb[o];                 gh -= f.hw * v1; gw -= f.hh * v1; unsafeAtomicAdd(gvb + o,                 f.hh * f.hw * tgv); }
                if (f.ok2) { v2 = vb[o + row];           gh -= f.lw * v2; gw += f.hh * v2; unsafeAtomicAdd(gvb + o + row,           f.hh * f.lw * tgv); }
                if (f.ok3) { v3 = vb[o + W * row];       gh += f.hw * v3; gw -= f.lh * v3; unsafeAtomicAdd(gvb + o + W * row,       f.lh * f.hw * tgv); }
                if (f.ok4) { v4 = vb[o + W * row + row]; gh += f.lw * v4; gw += f.lh * v4; unsafeAtomicAdd(gvb + o + W * row + row, f.lh * f.lw * tgv); }
                const T val = f.hh * f.hw * v1 + f.hh * f.lw * v2 + f.lh * f.hw * v3 + f.lh * f.lw * v4;
                unsafeAtomicAdd(gattp + l * P + p, top * val);                        // .cuh:231
                unsafeAtomicAdd(glocp + (l * P + p) * 2, static_cast<T>(W) * gw * tgv);      // .cuh:232
                unsafeAtomicAdd(glocp + (l * P + p) * 2 + 1, static_cast<T>(H) * gh * tgv);  // .cuh:233
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256)
void msda_indices_kernel(const int64_t *__restrict__ shapes, const T *__restrict__ loc, int32_t *__restrict__ idx,
                         int L, int P, int64_t n)
{
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int l = static_cast<int>((i / P) % L);
        const int H = static_cast<int>(shapes[2 * l]), W = static_cast<int>(shapes[2 * l + 1]);
        const Foot<T> f = footprint(pix_coord(loc[2 * i + 1], H), pix_coord(loc[2 * i], W), H, W);
        int4 o = make_int4(0, 0, 0, 0);
        if (f.inwin) o = make_int4(1, f.h_low, f.w_low, int(f.ok1) | (int(f.ok2) << 1) | (int(f.ok3) << 2) | (int(f.ok4) << 3));
        *reinterpret_cast<int4 *>(idx + 4 * i) = o;
    }
}

int grid_for(int64_t n, int block)
{
    const int64_t g = (n + block - 1) / block;
    return static_cast<int>(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

// how many 32-pair rounds each workgroup walks: keep >= ~8 workgroups per CU in flight on big
// problems, one round per workgroup on small ones (decoder: 4400 pairs / image)
int rounds_per_block(int B, int npairs)
{
    const int64_t rounds = static_cast<int64_t>(B) * ((npairs + 31) / 32);
    int it = static_cast<int>(rounds / (256 * 16));
    return it < 1 ? 1 : (it > 4 ? 4 : it);
}

}  // namespace

bool msda_fast_path(int dtype, int D, int L, int P)
{
    return dtype == 0 && D == 32 && L >= 1 && P >= 1 && (L * P) % 4 == 0 && L * P <= 64;
}

hipError_t msda_forward_launch(int dtype, const void *value, const int64_t *shapes, const int64_t *lstart,
                               const void *loc, const void *attn, void *out,
                               int B, int S, int M, int D, int L, int Lq, int P, hipStream_t st)
{
    const int64_t n = static_cast<int64_t>(B) * Lq * M * D;
    if (n == 0) return hipSuccess;
    struct Scope { hipStream_t s; Scope(int Lq_, hipStream_t s_) : s(s_) { profile_begin(0, Lq_, s_); } ~Scope() { profile_end(s); } } scope(Lq, st);
    if (msda_fast_path(dtype, D, L, P)) {
        const int npairs = Lq * M, iters = rounds_per_block(B, npairs);
        const int chunks = (npairs + 32 * iters - 1) / (32 * iters);
        const dim3 grid(static_cast<unsigned>(B) * chunks), block(kWaves * 64);
        const size_t lds = sizeof(float) * kWaves * slab_floats(L * P);
        auto a = [&](auto kern) {
            hipLaunchKernelGGL(kern, grid, block, lds, st, static_cast<const float *>(value), shapes, lstart,
                               static_cast<const float *>(loc), static_cast<const float *>(attn),
                               static_cast<float *>(out), B, S, M, L, P, npairs, iters);
        };
        if (L == 4 && P == 4)
            hipLaunchKernelGGL((msda_fwd_rec<4, 4>), grid, block, 0, st, static_cast<const float *>(value), shapes, lstart,
                               static_cast<const float *>(loc), static_cast<const float *>(attn),
                               static_cast<float *>(out), B, S, M, npairs, iters);
        else a(msda_fwd_d32<0, 0>);                            // other (L, P) of the D = 32 fast path: the slab kernel with run-time L, P
    } else if (dtype == 0) {
        hipLaunchKernelGGL(msda_fwd_generic<float>, dim3(grid_for(n, 256)), dim3(256), 0, st,
                           static_cast<const float *>(value), shapes, lstart, static_cast<const float *>(loc),
                           static_cast<const float *>(attn), static_cast<float *>(out), S, M, D, L, Lq, P, n);
    } else {
        hipLaunchKernelGGL(msda_fwd_generic<double>, dim3(grid_for(n, 256)), dim3(256), 0, st,
                           static_cast<const double *>(value), shapes, lstart, static_cast<const double *>(loc),
                           static_cast<const double *>(attn), static_cast<double *>(out), S, M, D, L, Lq, P, n);
    }
    return hipGetLastError();
}

namespace {
__global__ __launch_bounds__(256)
void zero_fill_kernel(uint4 *__restrict__ body, int64_t n16, unsigned char *__restrict__ head, int head_bytes,
                      unsigned char *__restrict__ tail, int tail_bytes)
{
    const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    for (int64_t i = t; i < n16; i += stride) body[i] = make_uint4(0u, 0u, 0u, 0u);
    if (t < head_bytes) head[t] = 0;
    if (t < tail_bytes) tail[t] = 0;
}
}  // namespace

hipError_t zero_fill_launch(void *p, int64_t bytes, hipStream_t st)
{
    if (bytes <= 0) return hipSuccess;
    unsigned char *b = static_cast<unsigned char *>(p);
    int head = static_cast<int>((16 - (reinterpret_cast<uintptr_t>(b) & 15)) & 15);
    if (head > bytes) head = static_cast<int>(bytes);
    const int64_t n16 = (bytes - head) / 16;
    const int tail = static_cast<int>(bytes - head - n16 * 16);
    int64_t blocks = (n16 + 256 * 4 - 1) / (256 * 4);                // ~4 stores per thread
    blocks = blocks < 1 ? 1 : (blocks > 256 * 32 ? 256 * 32 : blocks);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st,
                       reinterpret_cast<uint4 *>(b + head), n16, b, head, b + head + n16 * 16, tail);
    return hipGetLastError();
}

hipError_t msda_backward_launch_ex(int dtype, const void *value, const int64_t *shapes, const int64_t *lstart,
                                   const void *loc, const void *attn, const void *grad_out,
                                   void *grad_value, void *grad_loc, void *grad_attn,
                                   int B, int S, int M, int D, int L, int Lq, int P,
                                   const int64_t *shapes_host, const int64_t *lstart_host,
                                   void *workspace, int64_t workspace_bytes, hipStream_t st)
{
    const size_t e = dtype == 0 ? 4 : 8;
    const int64_t nv = static_cast<int64_t>(B) * S * M * D, ns = static_cast<int64_t>(B) * Lq * M * L * P;
    hipError_t err;
    const int64_t n = static_cast<int64_t>(B) * Lq * M * D;
    // MDETR_TUNE="msda_bwd=fused (default) | tiled | atomic" (tests): which grad_value strategy the fast path takes
    // (the decoder's 550 queries on the atomic or the tiled form instead: 454 / 453 img/s against 466, profiles/r06r_)
    const int bwd_env = [] { char tb[16]; const char *ev = tune_str("msda_bwd", tb, sizeof(tb)); return !ev || !*ev || ev[0] == 'f' ? 0 : (ev[0] == 't' ? 1 : 2); }();
    const bool fast = msda_fast_path(dtype, D, L, P);
    if (fast && bwd_env == 0 && n && ns && shapes_host && lstart_host && workspace) {
        // one-pass backward (msda_fused.hip): writes all three outputs completely, no zero fill
        err = msda_backward_fused_launch(shapes_host, lstart_host, value, static_cast<const float *>(loc), static_cast<const float *>(attn),
                                         grad_out, static_cast<float *>(grad_value), static_cast<float *>(grad_loc),
                                         static_cast<float *>(grad_attn), workspace, workspace_bytes, B, S, M, D, L, Lq, P, 0, st);
        if (err != hipErrorNotSupported) return err;
    }
    if (nv && (err = zero_fill_launch(grad_value, nv * e, st)) != hipSuccess) return err;
    if (n == 0 || ns == 0) return hipSuccess;
    if (!fast) {   // generic path accumulates grad_loc / grad_attn with atomics
        if ((err = zero_fill_launch(grad_loc, ns * 2 * e, st)) != hipSuccess) return err;
        if ((err = zero_fill_launch(grad_attn, ns * e, st)) != hipSuccess) return err;
    }
    if (fast) {
        const int var_env = tune_int("msda_bwd_variant", -1);
        // tile-privatised grad_value (msda_tiled.hip) when the geometry qualifies; the kernel below then
        // only produces grad_loc / grad_attn (VAR 2)
        const bool try_tiled = shapes_host && lstart_host && workspace && var_env != 0 && var_env != 1 && bwd_env != 2 &&
                               msda_tiled_workspace_bytes(shapes_host, lstart_host, B, S, M, D, L, Lq, P) > 0 &&
                               msda_tiled_workspace_bytes(shapes_host, lstart_host, B, S, M, D, L, Lq, P) <= workspace_bytes;
        unsigned *absmax2 = try_tiled ? static_cast<unsigned *>(workspace) : nullptr;
        if (try_tiled && (err = zero_fill_launch(absmax2, 8, st)) != hipSuccess) return err;
        const int var = try_tiled ? 2 : (var_env == 0 ? 0 : 1);
        const int npairs = Lq * M, iters = rounds_per_block(B, npairs);
        const int chunks = (npairs + 32 * iters - 1) / (32 * iters);
        const dim3 grid(static_cast<unsigned>(B) * chunks), block(kWaves * 64);
        const size_t lds = sizeof(float) * kWaves * slab_floats(L * P);
        auto a = [&](auto kern) {
            hipLaunchKernelGGL(kern, grid, block, lds, st, static_cast<const float *>(value), shapes, lstart,
                               static_cast<const float *>(loc), static_cast<const float *>(attn),
                               static_cast<const float *>(grad_out), static_cast<float *>(grad_value),
                               static_cast<float *>(grad_loc), static_cast<float *>(grad_attn),
                               B, S, M, L, P, npairs, iters, absmax2);
        };
        profile_begin(1, Lq, st);
        if (L == 4 && P == 4) {
            if (var == 0) a(msda_bwd_d32<4, 4, 0>); else if (var == 1) a(msda_bwd_d32<4, 4, 1>); else a(msda_bwd_d32<4, 4, 2>);
        } else {
            if (var == 2) a(msda_bwd_d32<0, 0, 2>); else a(msda_bwd_d32<0, 0, 1>);
        }
        profile_end(st);
        if (try_tiled) {
            // grad_value: tile-privatised scatter (msda_tiled.hip); the gather kernel above produced
            // grad_loc / grad_attn and max|grad_out|, max|attn| for the fixed-point scale
            err = msda_tiled_grad_value_launch(shapes_host, lstart_host, static_cast<const float *>(loc),
                                               static_cast<const float *>(attn), static_cast<const float *>(grad_out),
                                               static_cast<float *>(grad_value), workspace, workspace_bytes,
                                               B, S, M, D, L, Lq, P, /*absmax_ready=*/true, st);
            if (err != hipSuccess) return err;
        }
    } else if (dtype == 0) {
        profile_begin(1, Lq, st);
        hipLaunchKernelGGL(msda_bwd_generic<float>, dim3(grid_for(n, 256)), dim3(256), 0, st,
                           static_cast<const float *>(value), shapes, lstart, static_cast<const float *>(loc),
                           static_cast<const float *>(attn), static_cast<const float *>(grad_out),
                           static_cast<float *>(grad_value), static_cast<float *>(grad_loc),
                           static_cast<float *>(grad_attn), S, M, D, L, Lq, P, n);
        profile_end(st);
    } else {
        hipLaunchKernelGGL(msda_bwd_generic<double>, dim3(grid_for(n, 256)), dim3(256), 0, st,
                           static_cast<const double *>(value), shapes, lstart, static_cast<const double *>(loc),
                           static_cast<const double *>(attn), static_cast<const double *>(grad_out),
                           static_cast<double *>(grad_value), static_cast<double *>(grad_loc),
                           static_cast<double *>(grad_attn), S, M, D, L, Lq, P, n);
    }
    return hipGetLastError();
}

// ---- mixed-precision operator (bf16 value / out / grad_out, fp32 everything else): D = 32, L = P = 4 ------------
hipError_t msda_forward_bf16_launch(const void *value, const int64_t *shapes, const int64_t *lstart,
                                    const float *loc, const float *attn, void *out,
                                    int B, int S, int M, int D, int L, int Lq, int P, hipStream_t st)
{
    if (D != 32 || L != 4 || P != 4) return hipErrorNotSupported;
    if (static_cast<int64_t>(B) * Lq * M == 0) return hipSuccess;
    struct Scope { hipStream_t s; Scope(int Lq_, hipStream_t s_) : s(s_) { profile_begin(0, Lq_, s_); } ~Scope() { profile_end(s); } } scope(Lq, st);
    // (Round 4 measured a head-blocked form -- one head of one image per workgroup, levels 2 / 3 resident in LDS, four lanes
    // per query with 16-byte requests: 0.191-0.201 ms against this kernel's 0.193-0.196 at the encoder shape, 0.223 vs 0.202 for
    // N(0, 4 px) offsets, profiles/r04m_msda_fwd_head_blocked_vs_record.log.  Halving the rows gathered through memory bought
    // nothing: with one head per workgroup a 128-byte line carries one useful 64-byte row instead of two, so the number of L1
    // line look-ups per (query, head) pair is the same 32.  Not kept.  A second form blocked like the backward pass -- one head, one
    // region of 16 x 24 level-0 cells' queries per workgroup, EVERY level's reachable window of value rows staged in LDS (134 KB),
    // DPP quad broadcasts instead of LDS records -- measured 0.232-0.256 ms against 0.189 (0.274 vs 0.195 at N(0, 4 px)),
    // profiles/r04fw_msda_fwd_lds_windows_vs_record.log: one 16-wave workgroup per CU with two passes of 16 queries per wave behind a
    // 134 KB staging phase, 160 VGPRs wanted at a cap of 128.  Not kept either: this kernel stays the forward path.)
    const int npairs = Lq * M, iters = rounds_per_block(B, npairs);
    const int chunks = (npairs + 32 * iters - 1) / (32 * iters);
    hipLaunchKernelGGL((msda_fwd_rec<4, 4, __hip_bfloat16, __hip_bfloat16>), dim3(static_cast<unsigned>(B) * chunks), dim3(kWaves * 64), 0, st,
                       static_cast<const __hip_bfloat16 *>(value), shapes, lstart, loc, attn,
                       static_cast<__hip_bfloat16 *>(out), B, S, M, npairs, iters);
    return hipGetLastError();
}

hipError_t msda_backward_bf16_launch(const void *value, const int64_t *shapes, const int64_t *lstart,
                                     const float *loc, const float *attn, const void *grad_out,
                                     float *grad_value, float *grad_loc, float *grad_attn,
                                     int B, int S, int M, int D, int L, int Lq, int P,
                                     const int64_t *shapes_host, const int64_t *lstart_host,
                                     void *workspace, int64_t workspace_bytes, hipStream_t st)
{
    if (D != 32 || L != 4 || P != 4) return hipErrorNotSupported;
    const int64_t nv = static_cast<int64_t>(B) * S * M * D;
    hipError_t err;
    const int bwd_env = [] { char tb[16]; const char *ev = tune_str("msda_bwd", tb, sizeof(tb)); return !ev || !*ev || ev[0] == 'f' ? 0 : (ev[0] == 't' ? 1 : 2); }();
    if (bwd_env == 0 && static_cast<int64_t>(B) * Lq * M && shapes_host && lstart_host && workspace) {
        err = msda_backward_fused_launch(shapes_host, lstart_host, value, loc, attn, grad_out, grad_value, grad_loc, grad_attn,
                                         workspace, workspace_bytes, B, S, M, D, L, Lq, P, 2, st);
        if (err != hipErrorNotSupported) return err;
    }
    if (nv && (err = zero_fill_launch(grad_value, nv * 4, st)) != hipSuccess) return err;
    if (static_cast<int64_t>(B) * Lq * M == 0) return hipSuccess;
    const bool try_tiled = shapes_host && lstart_host && workspace && bwd_env != 2 &&
                           msda_tiled_workspace_bytes(shapes_host, lstart_host, B, S, M, D, L, Lq, P) > 0 &&
                           msda_tiled_workspace_bytes(shapes_host, lstart_host, B, S, M, D, L, Lq, P) <= workspace_bytes;
    unsigned *absmax2 = try_tiled ? static_cast<unsigned *>(workspace) : nullptr;
    if (try_tiled && (err = zero_fill_launch(absmax2, 8, st)) != hipSuccess) return err;
    const int npairs = Lq * M, iters = rounds_per_block(B, npairs);
    const int chunks = (npairs + 32 * iters - 1) / (32 * iters);
    const dim3 grid(static_cast<unsigned>(B) * chunks), block(kWaves * 64);
    const size_t lds = sizeof(float) * kWaves * slab_floats(L * P);
    auto a = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, block, lds, st, static_cast<const __hip_bfloat16 *>(value), shapes, lstart, loc, attn,
                           static_cast<const __hip_bfloat16 *>(grad_out), grad_value, grad_loc, grad_attn,
                           B, S, M, L, P, npairs, iters, absmax2);
    };
    profile_begin(1, Lq, st);
    if (try_tiled) a(msda_bwd_d32<4, 4, 2, __hip_bfloat16, __hip_bfloat16>);
    else a(msda_bwd_d32<4, 4, 1, __hip_bfloat16, __hip_bfloat16>);
    profile_end(st);
    if (try_tiled) {
        err = msda_tiled_grad_value_launch(shapes_host, lstart_host, loc, attn, grad_out, grad_value, workspace, workspace_bytes,
                                           B, S, M, D, L, Lq, P, /*absmax_ready=*/true, st, /*grad_out_dtype=*/2);
        if (err != hipSuccess) return err;
    }
    return hipGetLastError();
}

hipError_t msda_backward_launch(int dtype, const void *value, const int64_t *shapes, const int64_t *lstart,
                                const void *loc, const void *attn, const void *grad_out,
                                void *grad_value, void *grad_loc, void *grad_attn,
                                int B, int S, int M, int D, int L, int Lq, int P, hipStream_t st)
{
    return msda_backward_launch_ex(dtype, value, shapes, lstart, loc, attn, grad_out, grad_value, grad_loc, grad_attn,
                                   B, S, M, D, L, Lq, P, nullptr, nullptr, nullptr, 0, st);
}

hipError_t msda_indices_launch(int dtype, const int64_t *shapes, const void *loc, int32_t *idx,
                               int B, int M, int L, int Lq, int P, hipStream_t st)
{
    const int64_t n = static_cast<int64_t>(B) * Lq * M * L * P;
    if (n == 0) return hipSuccess;
    if (dtype == 0)
        hipLaunchKernelGGL(msda_indices_kernel<float>, dim3(grid_for(n, 256)), dim3(256), 0, st, shapes,
                           static_cast<const float *>(loc), idx, L, P, n);
    else
        hipLaunchKernelGGL(msda_indices_kernel<double>, dim3(grid_for(n, 256)), dim3(256), 0, st, shapes,
                           static_cast<const double *>(loc), idx, L, P, n);
    return hipGetLastError();
}

}  // namespace mdetr
