// monodetr_amd/csrc/conv_wgrad.hip -- weight gradient of the backbone's / pyramid's / depth head's 3x3 (stride 1 and 2) and
// strided 1x1 convolutions on the matrix cores: split-K over pixel tiles, both operands transposed on their way into LDS.
//
//   dW[n, t, e, c] = sum_{b, r, q} dY[b, r, q, n] * X[b, SI r + t - P, SI q + e - P, c]        (P = 1 for 3x3, 0 for 1x1)
//
// Reference: autograd of torchvision Bottleneck.conv2 / downsample (lib/models/monodetr/backbone.py:93-106), the 3x3 / stride-2
// pyramid level (monodetr.py:87-92) and the depth predictor's convolutions (depth_predictor.py:29-56), which the reference
// leaves to cuDNN.  MIOpen's igemm_wrw kernels take 71 us per 18.1-GFLOP layer here (255 TFLOP/s, profiles/r02v).
//
// The contraction index is the PIXEL, the slow axis of both channels-last operands: an MFMA lane needs 8 consecutive
// contraction values of one channel, so both operands are transposed -- in registers, an 8 x 8 bf16 block per thread (8 loads of
// 16 bytes = 8 channels of 8 rows; 32 two-word permutes; 8 stores of 16 bytes = 8 rows of 8 channels) -- on their way into LDS:
//   dYt[n][q][8 rows],   Xt[c][x column][8 rows]                       (a lane's operand = one aligned 16-byte read)
// The group of 8 contraction values is 8 ROWS of one column: a tap's column shift e then moves the read by whole 16-byte
// slots (8 consecutive columns would be shifted by 2 bytes: misaligned), and the tap's ROW shift t is taken out of the kernel's
// inner structure altogether -- a workgroup owns ONE tap row t and loads its X rows already shifted.  With SI = 2 the X columns
// are stored de-interleaved (even columns, then odd), as in conv_taps.hip.
// Workgroup = 8 waves: (tap row t, 128 output channels n, 64 input channels c, a chunk of the pixel tiles); wave w owns
// n-block w & 3 and c-block w >> 2: TS accumulator tiles D[n][c] (32 x 32) per wave, one per tap column.  A pixel tile is
// 8 output rows x kCols (16) output columns of one image = kCols / 2 MFMA k-steps (2 columns x 8 rows each).  The next tile's global loads are
// issued before the current tile's products and transposed into the single LDS buffer after them.
// Output: per-chunk partial gradients P[chunk][n][t][e][c] in fp32; colsum.hip adds the chunks in a fixed order and rounds once
// into the parameter's dtype (deterministic, no atomics).
// Algorithmic bytes = 2 B (H W C + OH OW N) + 4 TR TS N C;  flops = 2 B OH OW TR TS C N.  MFMA-bound by design.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mdetr_wave.h>

#include "conv_wgrad.h"
#include "mdetr_tune.h"
#include "mdetr_transpose.h"
#include "msda.h"       // profile scopes

namespace mdetr {
namespace {

constexpr int kThreadsG = 512;
constexpr int kNB = 128, kCB = 64;       // output / input channels per workgroup
// Columns of a pixel tile: 16.  With 32 the two transposed operand tiles take 103 - 134 KB of LDS -- one workgroup (8 waves) per CU;
// with 16 they take 54 - 70 KB and two workgroups fit.  Measured (profiles/r03s_wgrad_cols{32,16_wgs256}.json): 3x3 weight
// gradient of layer2 / 3 / 4 46 / 51 / 82 us -> 43 / 42 / 63 us at the same ~256 workgroups (more workgroups = more fp32 chunk
// partials to write and sum: 512 is slower again).
#ifndef MDETR_CONV_WGRAD_COLS
#define MDETR_CONV_WGRAD_COLS 16
#endif
constexpr int kRows = 8, kCols = MDETR_CONV_WGRAD_COLS;     // output pixels of a tile (columns: 16 or 32)
constexpr int kDyStride = kCols + 1;     // 16-byte slots per n row of dYt: odd, so the 16 lanes of a b128 group fall on distinct slots

struct WgradGeom {
    ConvWgradDims d;
    int bands, ctiles, units, chunks, nblocks, cblocks;
};

template <int SI, int TS>
__global__ __launch_bounds__(kThreadsG)
void conv_wgrad_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy, float *__restrict__ part, const WgradGeom g)
{
    constexpr int P = TS == 3 ? 1 : 0;
    // x columns of a tile in LDS: SI = 1: 32 + TS - 1 consecutive; SI = 2, 3 taps: 33 even + 32 odd; SI = 2, 1 tap: the 32 even ones
    constexpr int XC = SI == 1 ? kCols + TS - 1 : (TS == 3 ? 2 * kCols + 1 : kCols);
    constexpr int XSTRIDE = XC | 1;                                          // odd number of 16-byte slots per channel row
    constexpr int EVEN = kCols + 1;                                          // SI = 2, 3 taps: slots [0, 33) even columns, [33, 65) odd
    MDETR_DYNAMIC_LDS(unsigned char, wgrad_smem);
    bf16x8 *dyt = reinterpret_cast<bf16x8 *>(wgrad_smem);                    // [kNB][kDyStride]
    bf16x8 *xt = dyt + kNB * kDyStride;                                      // [kCB][XSTRIDE]
    const ConvWgradDims &d = g.d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int nsub = wave & 3, csub = wave >> 2;
    int id = blockIdx.x;
    const int chunk = id % g.chunks; id /= g.chunks;
    const int t = id % TS; id /= TS;                                         // tap row (TR = TS)
    const int cblk = id % g.cblocks; const int nblk = id / g.cblocks;
    const int n0 = nblk * kNB, c0 = cblk * kCB;

    f32x16 acc[TS];
#pragma unroll
    for (int e = 0; e < TS; ++e)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[e][i] = 0.f;
    // the bias gradient of a token-wise linear layer rides along (d.DB): db[n] = sum over pixels of dY[.][n] is the product of the
    // dY^T operand already in LDS with a matrix of ones -- one more MFMA per k-step on the waves of the first input-channel block,
    // no second pass over dY (a separate column sum: two launches and 42 MB per 81 600-row layer)
    const bool want_db = TS == 1 && d.DB != 0 && cblk == 0 && csub == 0;       // wave-uniform
    f32x16 accb;
#pragma unroll
    for (int i = 0; i < 16; ++i) accb[i] = 0.f;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = static_cast<__bf16>(1.0f);

    // ---- loading tasks: one 8 x 8 block (8 rows x 8 channels) per lane.  A WAVE takes 16 columns x 4 channel pieces of one operand
    // (lane = 16 piece + column): the 16 lanes of a ds_write_b128 group then store 16 CONSECUTIVE slots of one channel row --
    // conflict-free (8 lanes = the 8 pieces of one pixel, the coalescing-friendly order, put 4 lanes on each of two bank quads:
    // 68 % of the LDS cycles were conflicts, profiles/r03b_pmc_conv.json) -- while a load instruction still covers 64 contiguous
    // bytes of each of its 16 pixels.
    constexpr int CGX = (XC + 15) / 16;                                      // column groups of the X window
    constexpr int WTASKS = 2 * CGX + 4 * (kCols / 16);                       // X: CGX x 2 piece groups; dY: (kCols / 16) x 4
    constexpr int kWavesG = kThreadsG / 64;
    constexpr int ROUNDS = (WTASKS + kWavesG - 1) / kWavesG;
    constexpr int PF = ROUNDS <= 2 ? 2 : 1;                                  // tiles in flight ahead of the products (registers: 32 per round and tile)
    bf16x8 stage[PF][ROUNDS][8];
    // Loads go through buffer resources: a lane outside the image (the convolution's padding, a ragged tile) passes an offset
    // beyond the tensor and receives zeros -- no branch, no zero fill.  Per-lane offset = the column part, scalar offset = the
    // (image, row) part; the tensors are below 2^31 bytes (checked by the launcher).
    const mdetr_rsrc xr = make_rsrc(x, static_cast<unsigned>(static_cast<int64_t>(d.B) * d.H * d.W * d.C * 2));
    const mdetr_rsrc yr = make_rsrc(dy, static_cast<unsigned>(static_cast<int64_t>(d.B) * d.OH * d.OW * d.N * 2));
    const int ci = lane & 15, pi = lane >> 4;
    // what a lane's tasks are does not depend on the tile: (column of the window / of the tile, channel byte offset) per round
    int t_rel[ROUNDS], t_ch[ROUNDS], t_dst[ROUNDS];                          // t_dst: first LDS slot (row of its piece's channel 0)
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        const int wt = wave + rd * kWavesG;                                  // wave-uniform
        if (wt < 2 * CGX) {
            const int slot = 16 * (wt >> 1) + ci, piece = 4 * (wt & 1) + pi;
            t_rel[rd] = slot >= XC ? -1 : (SI == 1 ? slot : (TS == 3 ? (slot < EVEN ? 2 * slot : 2 * (slot - EVEN) + 1) : 2 * slot));
            t_ch[rd] = (c0 + piece * 8) * 2;
            t_dst[rd] = piece * 8 * XSTRIDE + slot;
        } else {
            const int k = wt - 2 * CGX, col = 16 * (k >> 2) + ci, piece = 4 * (k & 3) + pi;
            t_rel[rd] = col;
            t_ch[rd] = (wt < WTASKS && n0 + piece * 8 < d.N) ? (n0 + piece * 8) * 2 : -1;
            t_dst[rd] = piece * 8 * kDyStride + col;
        }
    }
    auto fetch = [&](int u, bf16x8 (&st)[ROUNDS][8]) {                       // global loads of pixel tile u
        const int ct = u % g.ctiles; u /= g.ctiles;
        const int band = u % g.bands; const int b = u / g.bands;
        const int r0 = band * kRows, q0 = ct * kCols;
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int wt = wave + rd * kWavesG;
            if (wt < 2 * CGX) {
                const int col = SI * q0 + t_rel[rd] - P;
                const unsigned lane_off = (t_rel[rd] >= 0 && col >= 0 && col < d.W) ? static_cast<unsigned>(col * d.C * 2 + t_ch[rd]) : kRsrcOob;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = SI * (r0 + i) + t - P;                   // uniform
                    const bool ok = row >= 0 && row < d.H;
                    st[rd][i] = rsrc_load_bf16x8(xr, ok ? lane_off : kRsrcOob, ok ? static_cast<unsigned>((b * d.H + row) * d.W) * static_cast<unsigned>(d.C * 2) : 0u);
                }
            } else if (wt < WTASKS) {
                const int col = q0 + t_rel[rd];
                const unsigned lane_off = (col < d.OW && t_ch[rd] >= 0) ? static_cast<unsigned>(col * d.N * 2 + t_ch[rd]) : kRsrcOob;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = r0 + i;
                    const bool ok = row < d.OH;
                    st[rd][i] = rsrc_load_bf16x8(yr, ok ? lane_off : kRsrcOob, ok ? static_cast<unsigned>((b * d.OH + row) * d.OW) * static_cast<unsigned>(d.N * 2) : 0u);
                }
            }
        }
    };
    auto commit = [&](const bf16x8 (&st)[ROUNDS][8]) {                       // transpose the staged blocks into LDS
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int wt = wave + rd * kWavesG;
            if (wt >= WTASKS) continue;
            bf16x8 tr[8];
            transpose8x8(st[rd], tr);
            if (wt < 2 * CGX) {
                if (16 * (wt >> 1) + ci < XC) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) xt[t_dst[rd] + q * XSTRIDE] = tr[q];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) dyt[t_dst[rd] + q * kDyStride] = tr[q];
            }
        }
    };
    auto products = [&]() {
        const bf16x8 *ap = dyt + (nsub * 32 + l31) * kDyStride + half;
        const bf16x8 *bp = xt + (csub * 32 + l31) * XSTRIDE;
#pragma unroll 4
        for (int ks = 0; ks < kCols / 2; ++ks) {
            const int q = 2 * ks + half;                                     // this half-wave's column of the k-step
            const bf16x8 a = ap[2 * ks];
#pragma unroll
            for (int e = 0; e < TS; ++e) {
                int slot;
                if (SI == 1) slot = q + e;
                else if (TS == 3) slot = (e & 1) ? EVEN + q : q + (e >> 1);  // 2q + e: even -> q + e / 2, odd -> after the even ones
                else slot = q;
                acc[e] = mfma_bf16(a, bp[slot], acc[e]);                     // D[n][c] += dY^T[n][8 rows] X[8 rows][c]
            }
            if (want_db) accb = mfma_bf16(a, ones, accb);                    // every column: sum over the 16 pixels of the k-step
        }
    };

    // tile u's loads were issued PF steps before its products (registers: one set per tile in flight), one LDS buffer
    int u = chunk;
    if (u < g.units) fetch(u, stage[0]);
    if (PF == 2 && u + g.chunks < g.units) fetch(u + g.chunks, stage[PF - 1]);
    while (u < g.units) {
#pragma unroll
        for (int s_ = 0; s_ < PF; ++s_) {
            if (u >= g.units) break;
            __syncthreads();                                                 // the previous tile's LDS reads are done
            commit(stage[s_]);
            const int nu = u + PF * g.chunks;
            if (nu < g.units) fetch(nu, stage[s_]);                          // in flight during the next PF tiles' products
            __syncthreads();
            products();
            u += g.chunks;
        }
    }

    // ---- epilogue: acc[e] register r of lane l = D[n = (r & 3) + 8 (r >> 2) + 4 half][c = l & 31]
    const int c = c0 + csub * 32 + l31;
    float *pp = part + static_cast<int64_t>(chunk) * (static_cast<int64_t>(d.N) * TS * TS * d.C + (d.DB ? d.N : 0));
    if (want_db && l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + nsub * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (n < d.N) pp[static_cast<int64_t>(d.N) * d.C + n] = accb[r];
        }
    }
#pragma unroll
    for (int e = 0; e < TS; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + nsub * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (n < d.N) pp[((static_cast<int64_t>(n) * TS + t) * TS + e) * d.C + c] = acc[e][r];
        }
}

template <int SI, int TS>
hipError_t launch(const void *x, const void *dy, float *part, WgradGeom g, hipStream_t st)
{
    constexpr int XC = SI == 1 ? kCols + TS - 1 : (TS == 3 ? 2 * kCols + 1 : kCols);
    constexpr size_t lds = (static_cast<size_t>(kNB) * kDyStride + static_cast<size_t>(kCB) * (XC | 1)) * 16;
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    auto kern = conv_wgrad_kernel<SI, TS>;
    static bool attr_set[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = -1;
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    const int64_t blocks = static_cast<int64_t>(g.chunks) * TS * g.cblocks * g.nblocks;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(kThreadsG), lds, st, static_cast<const __bf16 *>(x),
                       static_cast<const __bf16 *>(dy), part, g);
    return hipGetLastError();
}

WgradGeom geometry(const ConvWgradDims &d)
{
    WgradGeom g;
    g.d = d;
    g.bands = (d.OH + kRows - 1) / kRows;
    g.ctiles = (d.OW + kCols - 1) / kCols;
    g.units = d.B * g.bands * g.ctiles;
    g.nblocks = (d.N + kNB - 1) / kNB;
    g.cblocks = d.C / kCB;
    // one workgroup per CU (104-134 KB of LDS): about 256 workgroups, each with at least one pixel tile
    const int base = d.K * g.nblocks * g.cblocks;
    int target = 256;
    char tune_buf[16];
    if (const char *ev = tune_str("conv_wgrad_wgs", tune_buf, sizeof(tune_buf))) {                    // A/B runs: workgroups to aim for
        const int f = atoi(ev);
        if (f >= 64 && f <= 8192) target = f;
    }
    int chunks = target / (base > 0 ? base : 1);
    if (chunks < 1) chunks = 1;
    if (chunks > g.units) chunks = g.units;
    g.chunks = chunks;
    return g;
}

}  // namespace

bool conv_wgrad_supported(const ConvWgradDims &d, const void *x, const void *dy)
{
    const auto al = [](const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    // (K = 1 / stride 1: the weight gradient of a token-wise linear layer, its [T, C] / [T, N] matrices viewed as a 1 x 8 x T / 8 image)
    const bool shape = (d.K == 3 && (d.SI == 1 || d.SI == 2)) || (d.K == 1 && (d.SI == 2 || d.SI == 1));
    return shape && d.B > 0 && d.H > 0 && d.W > 0 && d.OH > 0 && d.OW > 0 && d.C > 0 && d.C % 64 == 0 && d.N > 0 && d.N % 32 == 0 &&
           al(x, 16) && al(dy, 16) && static_cast<int64_t>(d.N) * d.K * d.K * d.C < (1ll << 28) &&
           static_cast<int64_t>(d.B) * d.H * d.W * d.C < (1ll << 30) && static_cast<int64_t>(d.B) * d.OH * d.OW * d.N < (1ll << 30);
}

int conv_wgrad_chunks(const ConvWgradDims &d) { return geometry(d).chunks; }

hipError_t conv_wgrad_launch(const void *x, const void *dy, float *part, const ConvWgradDims &d, hipStream_t st)
{
    const WgradGeom g = geometry(d);
    ProfileScope prof(d.K == 1 && d.SI == 1 ? 11 : 9, conv_mflop(static_cast<int64_t>(d.B) * d.OH * d.OW, static_cast<int64_t>(d.C) * d.N * d.K * d.K), st,
                      2.0 * d.B * d.OH * d.OW * d.C * d.N * d.K * d.K / 1e6,
                      (2.0 * d.B * (static_cast<double>(d.H) * d.W * d.C + static_cast<double>(d.OH) * d.OW * d.N) + 4.0 * g.chunks * d.K * d.K * d.N * d.C) / 1e3);
    if (d.K == 3) return d.SI == 1 ? launch<1, 3>(x, dy, part, g, st) : launch<2, 3>(x, dy, part, g, st);
    return d.SI == 1 ? launch<1, 1>(x, dy, part, g, st) : launch<2, 1>(x, dy, part, g, st);
}

}  // namespace mdetr
