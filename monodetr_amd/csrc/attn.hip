// monodetr_amd/csrc/attn.hip -- fused dense multi-head attention (head_dim = 32) on CDNA4 MFMA.
//
// Replaces, on MonoDETR's hot path, the three nn.MultiheadAttention cores
//   depth cross-attention        depthaware_transformer.py:456-459   Lq = 550 (50), Lk = 1920
//   depth-encoder self-attention depth_predictor/transformer.py:59    Lq = Lk = 1920
//   decoder (grouped) self-attn  depthaware_transformer.py:496        Lq = Lk = 50, batch = B * 11
// whose reference path materialises QK^T, softmax, dropout and the head-averaged weights in HBM.
// Here softmax(q k^T * scale + mask) v runs flash-style: scores never leave registers.
//
// Tiling (wave64, `v_mfma_f32_32x32x16_bf16`):
//   * forward / dQ: one wave owns 32 query rows of one (batch, head); a 256-thread workgroup = 4
//     waves = 128 queries sharing K/V tiles of 64 keys staged in LDS.  Products are issued
//     "transposed" (S^T = K Q^T, O^T = V^T P^T, dQ^T = K^T dS^T) so that a lane's column index is
//     its query: row max / row sum / rescale are lane-local plus ONE exchange with lane^32.
//   * dK/dV: one wave owns 32 keys, loops over query tiles (S = Q K^T, dP = dO V^T, dV^T = dO^T P,
//     dK^T = Q^T dS) -- a lane's column index is its key.
//   * The accumulator (C/D) layout of a 32x32 MFMA -- row = (r&3) + 8(r>>2) + 4(lane>>5), col =
//     lane&31 -- is fed straight back as the B operand of the next product: the contraction index
//     is then visited in a permuted ("virtual") order, and the A operand is read from a TRANSPOSED
//     LDS copy with the same permutation (two 8-byte reads per lane), so no register shuffles.
//   * bf16 inputs: one bf16 MFMA per product step.  fp32 inputs: every operand x is split into THREE
//     bf16 parts hi + mid + lo (mid = bf16(x - hi), lo = bf16(x - hi - mid): 3 x 8 significand bits =
//     the 24 of an fp32 number, the split is exact) and each product is evaluated as
//     hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid (six MFMAs; the dropped mid*lo, lo*mid and
//     lo*lo terms are <= 2^-23 relative to |a||b|, the size of one fp32 rounding) -- products with
//     fp32 accuracy at 6/16 of the cost of the f32-input MFMA.  Round 5's two-part split carried 16
//     significand bits (2^-16 per product): the self-attention q / k projection gradients, which pass
//     through the ill-conditioned softmax Jacobian, came out at 3e-3 .. 6e-3 of float64, above the
//     1e-3 bar of the fp32 path.  fp32 accumulation in both modes; exp2-domain online
//     softmax with the scale folded into Q (or K), dropout on the probabilities from a stateless
//     counter hash of (seed, b, h, q, k) so forward and backward regenerate the same mask.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mdetr_wave.h>

#include "attn.h"
#include "mdetr_tune.h"
#include "msda.h"

namespace mdetr {
namespace {

constexpr int kD = 32;             // head dim
constexpr int kTile = 64;          // rows of a staged K/V (or Q/dO) tile
constexpr int kRowPad = 40;        // bf16 per row of a row-major tile (80 B: 16-B aligned, spreads banks)
constexpr int kTPad = 68;          // bf16 per row of a transposed tile [d][64 rows + 4]: 34 dwords per row = 2 x odd, so the 32 lanes of a
                                   // ds_read_b64 group (and the 16 of a ds_read2_b64 group) hit 64 (32) distinct banks; 72 was 2-way (PMC: 52 % conflict cycles)
constexpr float kLog2e = 1.4426950408889634f;

struct AttnArgs {
    const void *q, *k, *v;         // [B, L, H*32] with row strides (elements); last dim contiguous
    const uint8_t *kpm;            // [B, Lk] key padding mask (nonzero = ignore) or null
    int B, H, Lq, Lk;
    int64_t q_bs, k_bs, v_bs;      // batch strides (elements)
    int q_rs, k_rs, v_rs;          // row strides (elements)
    float scale, dropout_p;
    uint64_t seed;
    const uint64_t *seed_dev;
};

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return mfma_bf16(a, b, c); }      // mdetr_wave.h

struct Frag { bf16x8 h, m, l; };    // operand fragment: hi part (+ mid and lo parts in split mode)

template <bool SP>
__device__ __forceinline__ f32x16 mfmaX(const Frag &a, const Frag &b, f32x16 c)
{
    c = mfma(a.h, b.h, c);
    if (SP) {                        // smallest terms first would be more accurate still; the order costs nothing either way
        c = mfma(a.h, b.m, c); c = mfma(a.m, b.h, c);
        c = mfma(a.h, b.l, c); c = mfma(a.l, b.h, c); c = mfma(a.m, b.m, c);
    }
    return c;
}

struct HiLo { __bf16 h, m, l; };
__device__ __forceinline__ HiLo split_bf16(float x)
{
    HiLo r;
    r.h = static_cast<__bf16>(x);
    const float r1 = x - static_cast<float>(r.h);                 // exact (Sterbenz-like: hi is x rounded to 8 bits)
    r.m = static_cast<__bf16>(r1);
    r.l = static_cast<__bf16>(r1 - static_cast<float>(r.m));      // exact again; what is left has <= 8 significant bits
    return r;
}

// raw v_exp_f32 (exp2f() adds denormal-range fix-ups that cost ~5 extra instructions per element;
// probabilities below 2^-126 may flush to zero here, which is immaterial for a softmax)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ f32x16 zero16()
{
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// row index (contraction / output row) of accumulator register r for this lane half
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Keep-probability test for dropout: stateless, a function of the element's coordinates, so that forward and both backward
// kernels regenerate the same mask.  An element costs ONE full-rate multiply: keep(q, k) = mul_u24(Qs, Ks) >= thresh, where
//   Qs = strong32(q * kDropQMul ^ f(seed, b, h)) | 1      per query  (its lane's constant in forward / dQ; a 64-entry LDS
//   Ks = strong32((k + c) * kDropKMul)                    per key     table per staged tile in the kernel that loops over it)
// are 32-bit finalised hashes (the murmur3 mixer: two 32-bit multiplies, paid once per row, not per element) and mul_u24 is the
// low 32 bits of the product of their low 24 bits -- multiplication by an odd number permutes the residues, so the product is
// uniform; it is non-linear in both terms, so no additive / xor relation ties the four elements of a (q, q', k, k') rectangle
// together (xor of the two hashes alone would: three dropped corners made the fourth 8 x likelier).  Measured on 17 M elements
// (tests/test_dropout_hash_cpu.py): drop rate 0.1000, correlations at 90 lags along q, k, head, image and the diagonals below
// 3.6 sigma of the noise level, binomial row / column / tile counts.
// Before (rounds 1-2) every element paid xor, shift, xor and a QUARTER-rate 32-bit multiply: 40 of the forward kernel's ~100
// VALU cycles per element at p = 0.1; the compare + select that remain are 8, the multiply 4.
// The seed enters Qs before the finaliser: masks of seeds that differ in one bit are independent (the one-multiply hash of
// rounds 1-2 gave -0.11 for seed, seed + 1; callers still step seeds by the 64-bit golden ratio, attn_ext._next_seed).
constexpr unsigned kDropQMul = 0x9E3779B1u, kDropKMul = 0x85EBCA77u;
__device__ __forceinline__ unsigned strong32(unsigned v)
{
    v ^= v >> 16; v *= 0x85EBCA6Bu; v ^= v >> 13; v *= 0xC2B2AE35u; v ^= v >> 16;
    return v;
}
// the part of the query term that does not depend on the query
__device__ __forceinline__ unsigned drop_qconst(uint64_t seed, int b, int h)
{
    return static_cast<unsigned>(seed) ^ (static_cast<unsigned>(seed >> 32) + static_cast<unsigned>(b * 131 + h) * 0xC2B2AE3Du);
}
__device__ __forceinline__ unsigned drop_qs(unsigned qconst, int q) { return strong32((static_cast<unsigned>(q) * kDropQMul) ^ qconst) | 1u; }
__device__ __forceinline__ unsigned drop_ks(int k) { return strong32((static_cast<unsigned>(k) + 0x7F4A7C15u) * kDropKMul); }
__device__ __forceinline__ bool keep_elem(unsigned qs, unsigned ks, unsigned thresh) { return __umul24(qs, ks) >= thresh; }

template <typename T> __device__ __forceinline__ void load8(const T *p, float (&o)[8]);
template <> __device__ __forceinline__ void load8<float>(const float *p, float (&o)[8])
{
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <> __device__ __forceinline__ void load8<__bf16>(const __bf16 *p, float (&o)[8])
{
    const bf16x8 v = *reinterpret_cast<const bf16x8 *>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = static_cast<float>(v[i]);
}

template <typename T> __device__ __forceinline__ void store4(T *p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float *p, float a, float b, float c, float d)
{
    *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store4<__bf16>(__bf16 *p, float a, float b, float c, float d)
{
    bf16x4 v;
    v[0] = static_cast<__bf16>(a); v[1] = static_cast<__bf16>(b); v[2] = static_cast<__bf16>(c); v[3] = static_cast<__bf16>(d);
    *reinterpret_cast<bf16x4 *>(p) = v;
}

// Stage rows [r0, r0+64) x 32 of a [L, row_stride] matrix into LDS: row-major bf16 copy `rm`
// ([64][kRowPad]) and/or transposed copy `tr` ([32][kTPad]); rows >= L are zero.  In split mode the
// mid / lo parts go to rm + {1, 2} * kRmSize / tr + {1, 2} * kTrSize.  256 threads.
constexpr int kRmSize = kTile * kRowPad, kTrSize = kD * kTPad;

// 8 consecutive elements exactly as loaded (conversion is deferred to the LDS store so that the
// global load of the NEXT tile stays in flight across the compute on the current one)
template <typename T> struct Raw8;
template <> struct Raw8<float> { float4 a, b; };
template <> struct Raw8<__bf16> { bf16x8 v; };

__device__ __forceinline__ void raw_zero(Raw8<float> &r) { r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.b = r.a; }
__device__ __forceinline__ void raw_zero(Raw8<__bf16> &r)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = static_cast<__bf16>(0.f);
}
__device__ __forceinline__ void raw_load(Raw8<float> &r, const float *p) { r.a = *reinterpret_cast<const float4 *>(p); r.b = *reinterpret_cast<const float4 *>(p + 4); }
__device__ __forceinline__ void raw_load(Raw8<__bf16> &r, const __bf16 *p) { r.v = *reinterpret_cast<const bf16x8 *>(p); }
__device__ __forceinline__ void raw_floats(const Raw8<float> &r, float (&o)[8])
{
    o[0] = r.a.x; o[1] = r.a.y; o[2] = r.a.z; o[3] = r.a.w; o[4] = r.b.x; o[5] = r.b.y; o[6] = r.b.z; o[7] = r.b.w;
}
__device__ __forceinline__ void raw_floats(const Raw8<__bf16> &r, float (&o)[8])
{
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = static_cast<float>(r.v[i]);
}

// Staging map: which (row, 8-element chunk) of the 64 x 32 tile thread t loads and stores.
//   default : row = t / 4, chunk = t % 4 (four consecutive threads cover a row's 64 bytes);
//   remap   : row = (t & 15) + 16 (t >> 6), chunk = (t >> 4) & 3 -- the same 64 chunks per wave-level load (the
//             global side is unchanged), but inside a 32-lane store group only two chunk values occur, 16 banks
//             apart, so both the transposing 2-byte stores and the 16-byte row-major stores become conflict-free
//             in the LDS bank model (tests/test_lds_bank_model_cpu.py; they are 2-way today, the remaining 23 % of
//             conflict cycles).  Compile with -DMDETR_ATTN_STAGE_REMAP=1 to try it: off until it has run on a GPU.
#ifndef MDETR_ATTN_STAGE_REMAP
#define MDETR_ATTN_STAGE_REMAP 0
#endif

// this thread's 8 elements of tile rows [r0, r0+64): row = tid/4, columns (tid%4)*8 ..
template <typename T>
__device__ __forceinline__ Raw8<T> tile_load(const T *base, int row_stride, int r0, int L)
{
#if MDETR_ATTN_STAGE_REMAP
    const int t = threadIdx.x & 255;                             // (a workgroup may hold two 256-thread staging groups: key split)
    const int row = (t & 15) + 16 * (t >> 6), dc = ((t >> 4) & 3) * 8;
#else
    const int t = threadIdx.x & 255;                             // (a workgroup may hold two 256-thread staging groups: key split)
    const int row = t >> 2, dc = (t & 3) * 8;
#endif
    Raw8<T> r;
    if (r0 + row < L) raw_load(r, base + static_cast<int64_t>(r0 + row) * row_stride + dc);
    else raw_zero(r);
    return r;
}

template <typename T, bool RM, bool TR, bool SP>
__device__ __forceinline__ void tile_store(const Raw8<T> &raw, __bf16 *rm, __bf16 *tr)
{
#if MDETR_ATTN_STAGE_REMAP
    const int t = threadIdx.x & 255, row = (t & 15) + 16 * (t >> 6), dc = ((t >> 4) & 3) * 8;
#else
    const int t = threadIdx.x & 255, row = t >> 2, dc = (t & 3) * 8;
#endif
    float x[8];
    raw_floats(raw, x);
    bf16x8 vh, vm, vl;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (SP) { const HiLo s2 = split_bf16(x[i]); vh[i] = s2.h; vm[i] = s2.m; vl[i] = s2.l; }
        else vh[i] = static_cast<__bf16>(x[i]);
    }
    if (RM) {
        *reinterpret_cast<bf16x8 *>(rm + row * kRowPad + dc) = vh;
        if (SP) {
            *reinterpret_cast<bf16x8 *>(rm + kRmSize + row * kRowPad + dc) = vm;
            *reinterpret_cast<bf16x8 *>(rm + 2 * kRmSize + row * kRowPad + dc) = vl;
        }
    }
    if (TR) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            tr[(dc + i) * kTPad + row] = vh[i];
            if (SP) { tr[kTrSize + (dc + i) * kTPad + row] = vm[i]; tr[2 * kTrSize + (dc + i) * kTPad + row] = vl[i]; }
        }
    }
}

// A operand from a row-major tile: lane's row = sub*32 + (lane&31), 8 contiguous columns
template <bool SP>
__device__ __forceinline__ Frag frag_rows(const __bf16 *rm, int sub, int kstep, int lane)
{
    const int off = (sub * 32 + (lane & 31)) * kRowPad + kstep * 16 + (lane >> 5) * 8;
    Frag f;
    f.h = *reinterpret_cast<const bf16x8 *>(rm + off);
    if (SP) { f.m = *reinterpret_cast<const bf16x8 *>(rm + kRmSize + off); f.l = *reinterpret_cast<const bf16x8 *>(rm + 2 * kRmSize + off); }
    return f;
}

__device__ __forceinline__ bf16x8 read_cols(const __bf16 *p)
{
    const bf16x4 lo = *reinterpret_cast<const bf16x4 *>(p), hi = *reinterpret_cast<const bf16x4 *>(p + 8);
    bf16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return v;
}

// A operand from a transposed tile [d][rows]: lane's row = d = lane&31; contraction over the tile's
// rows in the accumulator ("virtual") order: rows base+0..3 and base+8..11, base = sub*32+16*kstep+4*half
template <bool SP>
__device__ __forceinline__ Frag frag_cols(const __bf16 *tr, int sub, int kstep, int lane)
{
    const int off = (lane & 31) * kTPad + sub * 32 + kstep * 16 + (lane >> 5) * 4;
    Frag f;
    f.h = read_cols(tr + off);
    if (SP) { f.m = read_cols(tr + kTrSize + off); f.l = read_cols(tr + 2 * kTrSize + off); }
    return f;
}

// B operand from accumulator registers: elements 8*kstep .. 8*kstep+7
template <bool SP>
__device__ __forceinline__ Frag frag_acc(const float (&p)[16], int kstep)
{
    Frag f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (SP) { const HiLo s2 = split_bf16(p[8 * kstep + i]); f.h[i] = s2.h; f.m[i] = s2.m; f.l[i] = s2.l; }
        else f.h[i] = static_cast<__bf16>(p[8 * kstep + i]);
    }
    return f;
}

// B operand from global rows owned by lanes: row = own row (lane&31), columns 16*kstep + 8*half + 0..7
template <typename T, bool SP>
__device__ __forceinline__ void own_row_frags(const T *rowp, bool valid, float mul, int lane, Frag (&f)[2])
{
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float x[8];
        if (valid) load8<T>(rowp + ks * 16 + (lane >> 5) * 8, x);
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (SP) { const HiLo s2 = split_bf16(x[i] * mul); f[ks].h[i] = s2.h; f[ks].m[i] = s2.m; f[ks].l[i] = s2.l; }
            else f[ks].h[i] = static_cast<__bf16>(x[i] * mul);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// DROP: dropout on the probabilities, a compile-time property of the kernel -- as a run-time flag it put a scalar branch
// around every element's hash (16 per sub-tile), which kept the exponentials, the hash and the conversions of
// neighbouring elements from being scheduled together.  The 1 / (1 - p) rescale is applied once to the output.
// KS: key split (an experiment, off by default: see key_split()).  With few queries (the decoder's 550: 5 query tiles x 8 heads x
// 8 images = 320 workgroups, 1 280 waves for 1 024 SIMDs) a SIMD holds about one wave.  KS = 2 puts two 4-wave groups into a workgroup: both own
// the SAME 128 queries, each walks half of the key tiles with its own staging buffers, and the two partial results
// (running maximum, sum, accumulator) are merged through LDS at the end -- twice the waves, no second launch.
template <typename T, bool DROP, int KS>
__global__ __launch_bounds__(256 * KS)
void attn_fwd_kernel(const AttnArgs a, T *__restrict__ out, float *__restrict__ lse2)
{
    constexpr bool SP = sizeof(T) == 4;                  // fp32 I/O: hi / mid / lo split operands
    constexpr int SPC = SP ? 3 : 1;
    __shared__ __attribute__((aligned(16))) __bf16 Ks_all[KS * SPC * kRmSize];
    __shared__ __attribute__((aligned(16))) __bf16 Vt_all[KS * SPC * kTrSize];
    __shared__ __attribute__((aligned(16))) unsigned kh_all[KS * kTile];  // dropout: the staged keys' hashed terms
    __shared__ float merge[KS == 2 ? 4 * 64 * 18 : 1];                    // KS = 2: group 1's (acc[16], m, l) per lane
    const int part = KS == 2 ? static_cast<int>(threadIdx.x >> 8) : 0, tl = threadIdx.x & 255;
    __bf16 *Ks = Ks_all + part * SPC * kRmSize, *Vt = Vt_all + part * SPC * kTrSize;
    unsigned *kh = kh_all + part * kTile;
    const int lane = tl & 63, wave = tl >> 6, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const bool qv = q < a.Lq;
    const T *Q = static_cast<const T *>(a.q) + b * a.q_bs + h * kD;
    const T *K = static_cast<const T *>(a.k) + b * a.k_bs + h * kD;
    const T *V = static_cast<const T *>(a.v) + b * a.v_bs + h * kD;

    Frag qf[2];
    own_row_frags<T, SP>(Q + static_cast<int64_t>(q) * a.q_rs, qv, a.scale * kLog2e, lane, qf);

    const uint64_t seed_eff = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
    const unsigned qs = drop_qs(drop_qconst(seed_eff, b, h), q);   // this lane's query: constant over the key loop
    const unsigned thresh = DROP ? static_cast<unsigned>(a.dropout_p * 4294967296.0) : 0u;
    const float rinv = DROP ? 1.f / (1.f - a.dropout_p) : 1.f;

    float m = -__builtin_inff(), l = 0.f;
    f32x16 acc = zero16();

    // every group walks `trips` key tiles (the same count: the barriers are the workgroup's); a tile past the end is all padding
    const int ntiles = (a.Lk + kTile - 1) / kTile, trips = (ntiles + KS - 1) / KS;
    const int kfirst = part * trips * kTile;
    Raw8<T> rk = tile_load<T>(K, a.k_rs, kfirst, a.Lk), rv = tile_load<T>(V, a.v_rs, kfirst, a.Lk);
    for (int it = 0; it < trips; ++it) {
        const int k0 = kfirst + it * kTile;
        __syncthreads();                                         // everyone is done reading the previous tile
        tile_store<T, true, false, SP>(rk, Ks, nullptr);
        tile_store<T, false, true, SP>(rv, nullptr, Vt);
        if (DROP && tl < kTile) kh[tl] = drop_ks(k0 + tl);
        __syncthreads();
        if (it + 1 < trips) {                                    // next tile's loads fly during this tile's math
            rk = tile_load<T>(K, a.k_rs, k0 + kTile, a.Lk);
            rv = tile_load<T>(V, a.v_rs, k0 + kTile, a.Lk);
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (k0 + sub * 32 >= a.Lk) break;                    // block-uniform
            f32x16 s = zero16();
            s = mfmaX<SP>(frag_rows<SP>(Ks, sub, 0, lane), qf[0], s);     // S^T[key][query]
            s = mfmaX<SP>(frag_rows<SP>(Ks, sub, 1, lane), qf[1], s);
            float p[16];
            float mx = m;
            if (!a.kpm && k0 + sub * 32 + 32 <= a.Lk) {              // block-uniform: whole sub-tile valid, no mask
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r] = s[r]; mx = fmaxf(mx, p[r]); }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + sub * 32 + acc_row(r, half);
                    bool ok = key < a.Lk;
                    if (a.kpm) ok = ok && (a.kpm[static_cast<int64_t>(b) * a.Lk + (ok ? key : 0)] == 0);
                    p[r] = ok ? s[r] : -__builtin_inff();
                    mx = fmaxf(mx, p[r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float msafe = mx == -__builtin_inff() ? 0.f : mx;
            const float alpha = fast_exp2(m - msafe);               // m = -inf -> 0
            const f32x2 ms2 = make_f32x2(msafe, msafe);
            f32x2 psum2 = make_f32x2(0.f, 0.f);
            const unsigned *khs = kh + sub * 32 + 4 * half;          // this lane's keys: rows acc_row(r, 0) from here (16-byte reads)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 d = sub2(make_f32x2(p[r], p[r + 1]), ms2);      // packed subtract: two elements per instruction
                f32x2 e = make_f32x2(fast_exp2(d.x), fast_exp2(d.y));
                psum2 = add2(psum2, e);
                if (DROP) {
                    e.x = keep_elem(qs, khs[acc_row(r, 0)], thresh) ? e.x : 0.f;
                    e.y = keep_elem(qs, khs[acc_row(r + 1, 0)], thresh) ? e.y : 0.f;
                }
                p[r] = e.x; p[r + 1] = e.y;
            }
            l = l * alpha + (psum2.x + psum2.y);
            if (__any(mx != m)) {                                    // wave-uniform: the running maximum moved for some query
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] *= alpha;
            }
            m = mx;
            acc = mfmaX<SP>(frag_cols<SP>(Vt, sub, 0, lane), frag_acc<SP>(p, 0), acc);   // O^T[d][query]
            acc = mfmaX<SP>(frag_cols<SP>(Vt, sub, 1, lane), frag_acc<SP>(p, 1), acc);
        }
    }
    l += __shfl_xor(l, 32);
    if (KS == 2) {                                               // merge the two key ranges: group 1 hands over, group 0 finishes
        float *mine = merge + (wave * 64 + lane) * 18;
        __syncthreads();
        if (part == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r] = acc[r];
            mine[16] = m; mine[17] = l;
        }
        __syncthreads();
        if (part == 1) return;
        const float m1 = mine[16], l1 = mine[17];
        const float mm = fmaxf(m, m1);
        const float a0 = m == -__builtin_inff() ? 0.f : fast_exp2(m - mm), a1 = m1 == -__builtin_inff() ? 0.f : fast_exp2(m1 - mm);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = acc[r] * a0 + mine[r] * a1;
        l = l * a0 + l1 * a1;
        m = mm;
    }
    if (qv) {
        const float inv = l > 0.f ? rinv / l : 0.f;              // fully masked row -> zeros; dropout's 1 / (1 - p) once per row
        T *o = out + (static_cast<int64_t>(b) * a.Lq + q) * (a.H * kD) + h * kD + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            store4<T>(o + 8 * g, acc[4 * g] * inv, acc[4 * g + 1] * inv, acc[4 * g + 2] * inv, acc[4 * g + 3] * inv);
        if (half == 0) lse2[(static_cast<int64_t>(b) * a.H + h) * a.Lq + q] = l > 0.f ? m + log2f(l) : -__builtin_inff();
    }
}

// ------------------------------------------------------------------------------------------------
// backward: D = rowsum(dO * O)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void attn_bwd_prep_kernel(const T *__restrict__ o, const T *__restrict__ d_o, float *__restrict__ dsum, int B, int H, int Lq)
{
    // one 8-lane group per (b, q, h): 32 channels as 8 x 4
    const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    const int64_t item = gid >> 3;
    const int c = static_cast<int>(gid & 7) * 4;
    float s = 0.f;
    const bool ok = item < static_cast<int64_t>(B) * Lq * H;
    if (ok) {
        const int64_t off = item * kD + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += static_cast<float>(o[off + i]) * static_cast<float>(d_o[off + i]);
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (ok && c == 0) {
        const int h = static_cast<int>(item % H);
        const int64_t bq = item / H;
        const int q = static_cast<int>(bq % Lq), b = static_cast<int>(bq / Lq);
        dsum[(static_cast<int64_t>(b) * H + h) * Lq + q] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// backward: dQ  (lane = query, loop over key tiles)
// ------------------------------------------------------------------------------------------------
template <typename T, bool DROP, int KS>
__global__ __launch_bounds__(256 * KS)
void attn_bwd_dq_kernel(const AttnArgs a, const T *__restrict__ d_o, const float *__restrict__ lse2,
                        const float *__restrict__ dsum, T *__restrict__ dq)
{
    constexpr bool SP = sizeof(T) == 4;
    constexpr int SPC = SP ? 3 : 1;
    __shared__ __attribute__((aligned(16))) __bf16 Ks_all[KS * SPC * kRmSize];
    __shared__ __attribute__((aligned(16))) __bf16 Vs_all[KS * SPC * kRmSize];
    __shared__ __attribute__((aligned(16))) __bf16 Kt_all[KS * SPC * kTrSize];
    __shared__ __attribute__((aligned(16))) unsigned kh_all[KS * kTile];  // dropout: the staged keys' hashed terms
    __shared__ float merge[KS == 2 ? 4 * 64 * 16 : 1];                    // KS = 2 (see attn_fwd_kernel): group 1's accumulator
    const int part = KS == 2 ? static_cast<int>(threadIdx.x >> 8) : 0, tl = threadIdx.x & 255;
    __bf16 *Ks = Ks_all + part * SPC * kRmSize, *Vs = Vs_all + part * SPC * kRmSize, *Kt = Kt_all + part * SPC * kTrSize;
    unsigned *kh = kh_all + part * kTile;
    const int lane = tl & 63, wave = tl >> 6, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const bool qv = q < a.Lq;
    const T *Q = static_cast<const T *>(a.q) + b * a.q_bs + h * kD;
    const T *K = static_cast<const T *>(a.k) + b * a.k_bs + h * kD;
    const T *V = static_cast<const T *>(a.v) + b * a.v_bs + h * kD;
    const int64_t orow = (static_cast<int64_t>(b) * a.Lq + q) * (a.H * kD) + h * kD;

    Frag qf[2], dof[2];
    own_row_frags<T, SP>(Q + static_cast<int64_t>(q) * a.q_rs, qv, a.scale * kLog2e, lane, qf);
    own_row_frags<T, SP>(d_o + orow, qv, 1.f, lane, dof);
    const int64_t stat = (static_cast<int64_t>(b) * a.H + h) * a.Lq + q;
    // a fully masked row has lse = -inf: read as +inf, exp2(s - inf) = 0 is its (zero) probability without a select per element
    float L2 = qv ? lse2[stat] : 0.f;
    if (L2 == -__builtin_inff()) L2 = __builtin_inff();
    const float Dq = qv ? dsum[stat] : 0.f;

    const uint64_t seed_eff = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
    const unsigned qs = drop_qs(drop_qconst(seed_eff, b, h), q);   // this lane's query: constant over the key loop
    const unsigned thresh = DROP ? static_cast<unsigned>(a.dropout_p * 4294967296.0) : 0u;
    const float rinv = DROP ? 1.f / (1.f - a.dropout_p) : 1.f;
    f32x16 acc = zero16();

    const int ntiles = (a.Lk + kTile - 1) / kTile, trips = (ntiles + KS - 1) / KS;
    const int kfirst = part * trips * kTile;
    Raw8<T> rk = tile_load<T>(K, a.k_rs, kfirst, a.Lk), rv = tile_load<T>(V, a.v_rs, kfirst, a.Lk);
    for (int it = 0; it < trips; ++it) {
        const int k0 = kfirst + it * kTile;
        __syncthreads();
        tile_store<T, true, true, SP>(rk, Ks, Kt);
        tile_store<T, true, false, SP>(rv, Vs, nullptr);
        if (DROP && tl < kTile) kh[tl] = drop_ks(k0 + tl);
        __syncthreads();
        if (it + 1 < trips) {
            rk = tile_load<T>(K, a.k_rs, k0 + kTile, a.Lk);
            rv = tile_load<T>(V, a.v_rs, k0 + kTile, a.Lk);
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (k0 + sub * 32 >= a.Lk) break;
            f32x16 s = zero16(), dp = zero16();
            s = mfmaX<SP>(frag_rows<SP>(Ks, sub, 0, lane), qf[0], s);
            s = mfmaX<SP>(frag_rows<SP>(Ks, sub, 1, lane), qf[1], s);
            dp = mfmaX<SP>(frag_rows<SP>(Vs, sub, 0, lane), dof[0], dp);  // dP^T[key][query] = V dO^T
            dp = mfmaX<SP>(frag_rows<SP>(Vs, sub, 1, lane), dof[1], dp);
            float ds[16];
            const bool plain = !a.kpm && k0 + sub * 32 + 32 <= a.Lk;   // block-uniform: whole sub-tile valid, no mask
            const unsigned *khs = kh + sub * 32 + 4 * half;
            const f32x2 l22 = make_f32x2(L2, L2);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 d = sub2(make_f32x2(s[r], s[r + 1]), l22);
                f32x2 p = make_f32x2(fast_exp2(d.x), fast_exp2(d.y));
                if (!plain) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int key = k0 + sub * 32 + acc_row(r + i, half);
                        bool ok = key < a.Lk;
                        if (a.kpm) ok = ok && (a.kpm[static_cast<int64_t>(b) * a.Lk + (ok ? key : 0)] == 0);
                        if (i == 0) p.x = ok ? p.x : 0.f; else p.y = ok ? p.y : 0.f;
                    }
                }
                f32x2 g = make_f32x2(dp[r], dp[r + 1]);
                if (DROP) {
                    g.x = keep_elem(qs, khs[acc_row(r, 0)], thresh) ? g.x : 0.f;
                    g.y = keep_elem(qs, khs[acc_row(r + 1, 0)], thresh) ? g.y : 0.f;
                }
                const f32x2 t = fma2(g, make_f32x2(rinv, rinv), make_f32x2(-Dq, -Dq));     // kept dP / (1 - p) - D
                const f32x2 dsv = mul2(p, t);
                ds[r] = dsv.x; ds[r + 1] = dsv.y;
            }
            acc = mfmaX<SP>(frag_cols<SP>(Kt, sub, 0, lane), frag_acc<SP>(ds, 0), acc);   // dQ^T[d][query]
            acc = mfmaX<SP>(frag_cols<SP>(Kt, sub, 1, lane), frag_acc<SP>(ds, 1), acc);
        }
    }
    if (KS == 2) {
        float *mine = merge + (wave * 64 + lane) * 16;
        __syncthreads();
        if (part == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[r] = acc[r];
        }
        __syncthreads();
        if (part == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += mine[r];
    }
    if (qv) {
        T *o = dq + orow + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            store4<T>(o + 8 * g, acc[4 * g] * a.scale, acc[4 * g + 1] * a.scale, acc[4 * g + 2] * a.scale, acc[4 * g + 3] * a.scale);
    }
}

// ------------------------------------------------------------------------------------------------
// backward: dK, dV  (lane = key, loop over query tiles)
// ------------------------------------------------------------------------------------------------
template <typename T, bool DROP>
__global__ __launch_bounds__(256)
void attn_bwd_dkv_kernel(const AttnArgs a, const T *__restrict__ d_o, const float *__restrict__ lse2,
                         const float *__restrict__ dsum, T *__restrict__ dk, T *__restrict__ dv)
{
    constexpr bool SP = sizeof(T) == 4;
    __shared__ __attribute__((aligned(16))) __bf16 Qs[(SP ? 3 : 1) * kRmSize];
    __shared__ __attribute__((aligned(16))) __bf16 Os[(SP ? 3 : 1) * kRmSize];
    __shared__ __attribute__((aligned(16))) __bf16 Qt[(SP ? 3 : 1) * kTrSize];
    __shared__ __attribute__((aligned(16))) __bf16 Ot[(SP ? 3 : 1) * kTrSize];
    __shared__ __attribute__((aligned(16))) float Ls[kTile];
    __shared__ __attribute__((aligned(16))) float Ds[kTile];
    __shared__ __attribute__((aligned(16))) unsigned qh[kTile];           // dropout: the staged queries' hashed terms
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int key = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const bool kv = key < a.Lk;
    const T *Q = static_cast<const T *>(a.q) + b * a.q_bs + h * kD;
    const T *K = static_cast<const T *>(a.k) + b * a.k_bs + h * kD;
    const T *V = static_cast<const T *>(a.v) + b * a.v_bs + h * kD;
    const T *dO = d_o + static_cast<int64_t>(b) * a.Lq * (a.H * kD) + h * kD;
    const float *L2b = lse2 + (static_cast<int64_t>(b) * a.H + h) * a.Lq;
    const float *Db = dsum + (static_cast<int64_t>(b) * a.H + h) * a.Lq;

    Frag kf[2], vf[2];
    own_row_frags<T, SP>(K + static_cast<int64_t>(key) * a.k_rs, kv, a.scale * kLog2e, lane, kf);
    own_row_frags<T, SP>(V + static_cast<int64_t>(key) * a.v_rs, kv, 1.f, lane, vf);
    bool key_ok = kv;
    if (a.kpm && kv) key_ok = a.kpm[static_cast<int64_t>(b) * a.Lk + key] == 0;

    const uint64_t seed_eff = a.seed + (a.seed_dev ? *a.seed_dev : 0ull);
    const unsigned ks = drop_ks(key);                             // this lane's key: constant over the query loop
    const unsigned qconst = drop_qconst(seed_eff, b, h);
    const unsigned thresh = DROP ? static_cast<unsigned>(a.dropout_p * 4294967296.0) : 0u;
    const float rinv = DROP ? 1.f / (1.f - a.dropout_p) : 1.f;
    f32x16 acck = zero16(), accv = zero16();

    Raw8<T> rq = tile_load<T>(Q, a.q_rs, 0, a.Lq), ro = tile_load<T>(dO, a.H * kD, 0, a.Lq);
    // rows past Lq and fully masked rows (lse = -inf) carry lse = +inf: exp2(s - inf) = 0 without a select per element
    float rl = __builtin_inff(), rd = 0.f;
    if (threadIdx.x < kTile && threadIdx.x < a.Lq) {
        rl = L2b[threadIdx.x]; rd = Db[threadIdx.x];
        if (rl == -__builtin_inff()) rl = __builtin_inff();
    }
    for (int q0 = 0; q0 < a.Lq; q0 += kTile) {
        __syncthreads();
        tile_store<T, true, true, SP>(rq, Qs, Qt);
        tile_store<T, true, true, SP>(ro, Os, Ot);
        if (threadIdx.x < kTile) {
            Ls[threadIdx.x] = rl; Ds[threadIdx.x] = rd;
            if (DROP) qh[threadIdx.x] = drop_qs(qconst, q0 + threadIdx.x);
        }
        __syncthreads();
        if (q0 + kTile < a.Lq) {
            rq = tile_load<T>(Q, a.q_rs, q0 + kTile, a.Lq);
            ro = tile_load<T>(dO, a.H * kD, q0 + kTile, a.Lq);
            if (threadIdx.x < kTile) {
                const int qq = q0 + kTile + threadIdx.x;
                rl = qq < a.Lq ? L2b[qq] : __builtin_inff();
                if (rl == -__builtin_inff()) rl = __builtin_inff();
                rd = qq < a.Lq ? Db[qq] : 0.f;
            }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (q0 + sub * 32 >= a.Lq) break;
            f32x16 s = zero16(), dp = zero16();
            s = mfmaX<SP>(frag_rows<SP>(Qs, sub, 0, lane), kf[0], s);     // S[query][key]
            s = mfmaX<SP>(frag_rows<SP>(Qs, sub, 1, lane), kf[1], s);
            dp = mfmaX<SP>(frag_rows<SP>(Os, sub, 0, lane), vf[0], dp);   // dP[query][key] = dO V^T
            dp = mfmaX<SP>(frag_rows<SP>(Os, sub, 1, lane), vf[1], dp);
            float pd[16], ds[16];
            const f32x2 ri2 = make_f32x2(rinv, rinv);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int qi = sub * 32 + acc_row(r, half);                      // (r even: qi + 1 is the row of register r + 1)
                const f32x2 d = sub2(make_f32x2(s[r], s[r + 1]), make_f32x2(Ls[qi], Ls[qi + 1]));
                f32x2 p = make_f32x2(fast_exp2(d.x), fast_exp2(d.y));
                if (!key_ok) p = make_f32x2(0.f, 0.f);                           // a padded / masked key (its lane's whole column)
                f32x2 g = make_f32x2(dp[r], dp[r + 1]), pk = p;
                if (DROP) {
                    const bool k0p = keep_elem(qh[qi], ks, thresh), k1p = keep_elem(qh[qi + 1], ks, thresh);
                    g.x = k0p ? g.x : 0.f; pk.x = k0p ? pk.x : 0.f;
                    g.y = k1p ? g.y : 0.f; pk.y = k1p ? pk.y : 0.f;
                }
                pd[r] = pk.x; pd[r + 1] = pk.y;                                  // (1 / (1 - p) is applied to dV once, at the end)
                const f32x2 t = sub2(mul2(g, ri2), make_f32x2(Ds[qi], Ds[qi + 1]));
                const f32x2 dsv = mul2(p, t);
                ds[r] = dsv.x; ds[r + 1] = dsv.y;
            }
            accv = mfmaX<SP>(frag_cols<SP>(Ot, sub, 0, lane), frag_acc<SP>(pd, 0), accv);   // dV^T[d][key] = dO^T P
            accv = mfmaX<SP>(frag_cols<SP>(Ot, sub, 1, lane), frag_acc<SP>(pd, 1), accv);
            acck = mfmaX<SP>(frag_cols<SP>(Qt, sub, 0, lane), frag_acc<SP>(ds, 0), acck);   // dK^T[d][key] = Q^T dS
            acck = mfmaX<SP>(frag_cols<SP>(Qt, sub, 1, lane), frag_acc<SP>(ds, 1), acck);
        }
    }
    if (kv) {
        const int64_t row = (static_cast<int64_t>(b) * a.Lk + key) * (a.H * kD) + h * kD + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            store4<T>(dk + row + 8 * g, acck[4 * g] * a.scale, acck[4 * g + 1] * a.scale, acck[4 * g + 2] * a.scale, acck[4 * g + 3] * a.scale);
            store4<T>(dv + row + 8 * g, accv[4 * g] * rinv, accv[4 * g + 1] * rinv, accv[4 * g + 2] * rinv, accv[4 * g + 3] * rinv);
        }
    }
}

AttnArgs make_args(const AttnProblem &p)
{
    AttnArgs a;
    a.q = p.q; a.k = p.k; a.v = p.v; a.kpm = p.key_padding_mask;
    a.B = p.B; a.H = p.H; a.Lq = p.Lq; a.Lk = p.Lk;
    a.q_bs = p.q_batch_stride; a.k_bs = p.k_batch_stride; a.v_bs = p.v_batch_stride;
    a.q_rs = p.q_row_stride; a.k_rs = p.k_row_stride; a.v_rs = p.v_row_stride;
    a.scale = p.scale; a.dropout_p = p.dropout_p; a.seed = p.seed; a.seed_dev = p.seed_dev;
    return a;
}

}  // namespace

// two key ranges per workgroup (KS = 2): bf16 only (the fp32 mode's split operands double the staging buffers), several key tiles
// to share.  OFF unless MDETR_TUNE="attn_ksplit=1" (tests): measured on the case it was written for (B = 8, 550 x 1920, dropout 0.1) the forward
// went from 0.056 to 0.061 ms and the backward did not move (profiles/r03z_attnbench_ks{0,auto}.json) -- the 8-wave workgroup's
// barriers and the merge cost more than the second wave per SIMD hides.
bool key_split(const AttnProblem &p)
{
    if (p.dtype == 0 || p.Lk < 4 * kTile) return false;
    char tune_buf[8];
    const char *ev = tune_str("attn_ksplit", tune_buf, sizeof(tune_buf));
    return ev && ev[0] == '1';
}

hipError_t attn_forward_launch(const AttnProblem &p, void *out, float *lse2, hipStream_t st)
{
    if (p.B == 0 || p.Lq == 0) return hipSuccess;
    const AttnArgs a = make_args(p);
    const dim3 grid((p.Lq + 127) / 128, p.H, p.B);
    const bool split = key_split(p);
    const dim3 block(split ? 512 : 256);
    profile_begin(4, p.Lq * 4096 + (p.Lk < 4096 ? p.Lk : 4095), st);
    const bool drop = p.dropout_p > 0.f;
    if (p.dtype == 0) {
        if (drop) hipLaunchKernelGGL((attn_fwd_kernel<float, true, 1>), grid, block, 0, st, a, static_cast<float *>(out), lse2);
        else hipLaunchKernelGGL((attn_fwd_kernel<float, false, 1>), grid, block, 0, st, a, static_cast<float *>(out), lse2);
    } else if (split) {
        if (drop) hipLaunchKernelGGL((attn_fwd_kernel<__bf16, true, 2>), grid, block, 0, st, a, static_cast<__bf16 *>(out), lse2);
        else hipLaunchKernelGGL((attn_fwd_kernel<__bf16, false, 2>), grid, block, 0, st, a, static_cast<__bf16 *>(out), lse2);
    } else {
        if (drop) hipLaunchKernelGGL((attn_fwd_kernel<__bf16, true, 1>), grid, block, 0, st, a, static_cast<__bf16 *>(out), lse2);
        else hipLaunchKernelGGL((attn_fwd_kernel<__bf16, false, 1>), grid, block, 0, st, a, static_cast<__bf16 *>(out), lse2);
    }
    profile_end(st);
    return hipGetLastError();
}

hipError_t attn_backward_launch(const AttnProblem &p, const void *out, const void *d_out, const float *lse2,
                                float *dsum, void *dq, void *dk, void *dv, hipStream_t st)
{
    if (p.B == 0) return hipSuccess;
    const AttnArgs a = make_args(p);
    const int64_t items = static_cast<int64_t>(p.B) * p.Lq * p.H;
    const dim3 gq((p.Lq + 127) / 128, p.H, p.B), gk((p.Lk + 127) / 128, p.H, p.B), block(256);
    const bool drop = p.dropout_p > 0.f;
    struct Scope { hipStream_t s; Scope(int key, hipStream_t s_) : s(s_) { profile_begin(5, key, s_); } ~Scope() { profile_end(s); } }
        scope(p.Lq * 4096 + (p.Lk < 4096 ? p.Lk : 4095), st);
    if (p.dtype == 0) {
        if (items) hipLaunchKernelGGL(attn_bwd_prep_kernel<float>, dim3(static_cast<unsigned>((items * 8 + 255) / 256)), block, 0, st,
                                      static_cast<const float *>(out), static_cast<const float *>(d_out), dsum, p.B, p.H, p.Lq);
        const float *go = static_cast<const float *>(d_out);
        if (p.Lq && drop) hipLaunchKernelGGL((attn_bwd_dq_kernel<float, true, 1>), gq, block, 0, st, a, go, lse2, dsum, static_cast<float *>(dq));
        if (p.Lq && !drop) hipLaunchKernelGGL((attn_bwd_dq_kernel<float, false, 1>), gq, block, 0, st, a, go, lse2, dsum, static_cast<float *>(dq));
        if (p.Lk && drop) hipLaunchKernelGGL((attn_bwd_dkv_kernel<float, true>), gk, block, 0, st, a, go, lse2, dsum, static_cast<float *>(dk), static_cast<float *>(dv));
        if (p.Lk && !drop) hipLaunchKernelGGL((attn_bwd_dkv_kernel<float, false>), gk, block, 0, st, a, go, lse2, dsum, static_cast<float *>(dk), static_cast<float *>(dv));
    } else {
        if (items) hipLaunchKernelGGL(attn_bwd_prep_kernel<__bf16>, dim3(static_cast<unsigned>((items * 8 + 255) / 256)), block, 0, st,
                                      static_cast<const __bf16 *>(out), static_cast<const __bf16 *>(d_out), dsum, p.B, p.H, p.Lq);
        const __bf16 *go = static_cast<const __bf16 *>(d_out);
        const bool split = key_split(p);
        const dim3 bq(split ? 512 : 256);
        if (p.Lq && drop && split) hipLaunchKernelGGL((attn_bwd_dq_kernel<__bf16, true, 2>), gq, bq, 0, st, a, go, lse2, dsum, static_cast<__bf16 *>(dq));
        if (p.Lq && !drop && split) hipLaunchKernelGGL((attn_bwd_dq_kernel<__bf16, false, 2>), gq, bq, 0, st, a, go, lse2, dsum, static_cast<__bf16 *>(dq));
        if (p.Lq && drop && !split) hipLaunchKernelGGL((attn_bwd_dq_kernel<__bf16, true, 1>), gq, bq, 0, st, a, go, lse2, dsum, static_cast<__bf16 *>(dq));
        if (p.Lq && !drop && !split) hipLaunchKernelGGL((attn_bwd_dq_kernel<__bf16, false, 1>), gq, bq, 0, st, a, go, lse2, dsum, static_cast<__bf16 *>(dq));
        if (p.Lk && drop) hipLaunchKernelGGL((attn_bwd_dkv_kernel<__bf16, true>), gk, block, 0, st, a, go, lse2, dsum, static_cast<__bf16 *>(dk), static_cast<__bf16 *>(dv));
        if (p.Lk && !drop) hipLaunchKernelGGL((attn_bwd_dkv_kernel<__bf16, false>), gk, block, 0, st, a, go, lse2, dsum, static_cast<__bf16 *>(dk), static_cast<__bf16 *>(dv));
    }
    return hipGetLastError();
}

}  // namespace mdetr
