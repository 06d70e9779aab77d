// monodetr_amd/csrc/conv_taps.hip -- the strided convolutions of the backbone and the feature pyramid, and the input gradients
// of those, as implicit GEMMs on the matrix cores with the im2col in LDS: one kernel over a rectangular set of TAPS.
//
//   y[b, r, c, n] = act( shift[n] + sum_{a < TR, e < TS, k} x[b, SI r + a - PT, SI c + e - PL, k] * w[n, a, e, k] )
//
// with the output pixel (r, c) written at an arbitrary (row, column) stride and offset.  What the ResNet-50 body, the fourth
// pyramid level and the depth predictor need of it (torchvision Bottleneck.conv2 / downsample of each stage's first block
// behind lib/models/monodetr/backbone.py:93-106; monodetr.py:87-92; depth_predictor.py:29-31):
//   * 3x3 / stride 2 / pad 1, forward:   TR = TS = 3, SI = 2, PT = PL = 1;
//   * 1x1 / stride 2, forward:           TR = TS = 1, SI = 2, PT = PL = 0;
//   * their INPUT GRADIENTS.  dX of a stride-2 convolution splits by the parity (pi, pj) of the input pixel: the pixels
//     (2r + pi, 2c + pj) form a stride-1 problem over the dY map, dX[2r + pi, 2c + pj] = sum over the taps t = 2a' + 1 - pi ...
//     i.e. one tap (t = 1) for an even coordinate and two (t = 2 reading dY[r], t = 0 reading dY[r + 1]) for an odd one:
//     four launches with 1x1, 1x2, 2x1 and 2x2 taps, SI = 1, PT = PL = 0, writing every second pixel of every second row.
//     (MIOpen's backward-data kernels for these take 48-64 us each in profiles/r02v; a zero-insertion form would do 4x the
//     work.)  The weight is addressed through (tap offset, tap step) pairs, so the four classes read ONE transposed,
//     tap-mirrored copy of it.
// Geometry as conv3x3.hip (whose fragment conventions, staging scheme and epilogue this file follows): a workgroup = 4 waves
// owns 4 output rows x 32 output columns x NB*32 output channels; wave i owns row i, lane & 31 a column; per 64-channel slab
// the input halo -- (SI*3 + TR) x (SI*31 + TS) pixels, zero outside the image -- is staged in LDS once and serves every tap;
// the weights of one tap row follow.  With SI = 2 the halo's columns are stored DE-INTERLEAVED (even columns, then odd): the
// 32 lanes of a tap then read 32 consecutive LDS rows instead of every second one (a 2-way bank conflict on every operand).
// Products transposed, Y^T[n][pixel], v_mfma_f32_32x32x16_bf16: shift, ReLU and the bf16 rounding in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mdetr_wave.h>

#include "conv_taps.h"
#include "mdetr_tune.h"
#include "msda.h"       // profile scopes

// Contraction channels per LDS slab.  32, not conv3x3.hip's 64: the stride-2 halo of a 4 x 32-pixel tile is 9 x 65 pixels -- 84 KB at
// 64 channels + padding, which with the weights left ONE workgroup (four waves) per CU; at 32 channels the tile is 47 + 31 KB and
// two workgroups fit.  Measured (profiles/r03q_strided_slab{64,32}.json): forward 3x3 / stride 2 of layer2 / 3 / 4 and the depth
// head 49 / 71 / 86 / 70 us -> 39 / 54 / 68 / 52 us; the 2048-channel pyramid level loses (125 -> 146 us: twice the stages).
#ifndef MDETR_CONV_TAPS_SLAB
#define MDETR_CONV_TAPS_SLAB 32
#endif

namespace mdetr {
namespace {

constexpr int kWavesT = 4;               // = output rows per workgroup
constexpr int kTileWT = 32;              // output columns per workgroup
constexpr int kSlabT = MDETR_CONV_TAPS_SLAB;   // contraction channels per LDS slab
constexpr int kPiecesT = kSlabT / 8;     // 16-byte pieces of a pixel's slab
constexpr int kPadT = kSlabT + 8;        // 40 bf16 = 20 dwords per LDS row (72 at 64 channels): the 16 lanes of a b128 group fall on 16 distinct bank quads

struct TapGeom {
    ConvTapsDims d;
    int tiles_x, tiles_y, tiles, ngroups, xcd_per;
    // split over the contraction channels (few output pixels against a long contraction: the fourth pyramid level, 2048 channels
    // x 9 taps into 8 x 6 x 20 pixels, kept 128 workgroups busy for 191 us): workgroup (split s, tile, group) contracts channel slabs
    // [s, s + 1) * C / ksplit and writes an fp32 partial [ksplit][B][OH][OW][N] (split 0 adds the shift); the caller sums the splits
    int ksplit = 1, inner_blocks = 0;
    float *part = nullptr;
};

template <int NB, bool RELU, int TR, int TS, int SI>
__device__ __forceinline__ void conv_taps_body(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w, const float *__restrict__ shift,
                                               __bf16 *__restrict__ y, const TapGeom &g, const int bid_)
{
    const int split = g.ksplit > 1 ? bid_ / g.inner_blocks : 0;               // uniform
    const int bid = g.ksplit > 1 ? bid_ - split * g.inner_blocks : bid_;
    constexpr int HH = SI * (kWavesT - 1) + TR;                              // halo rows
    constexpr int HWC = SI * (kTileWT - 1) + TS;                             // halo columns
    constexpr int PLANE = (HWC + 1) / 2;                                     // SI = 2: even columns [0, PLANE), odd columns after
    MDETR_DYNAMIC_LDS(unsigned char, taps_smem);
    __bf16 *halo = reinterpret_cast<__bf16 *>(taps_smem);                    // [HH][HWC][kPadT]
    __bf16 *wts = halo + HH * HWC * kPadT;                                   // [TS][NB*32][kPadT]
    float *shift_s = reinterpret_cast<float *>(wts + TS * NB * 32 * kPadT);  // [NB*32]
    const ConvTapsDims &d = g.d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    int group, t;
    if (g.xcd_per > 0) {                                                     // an XCD's L2 sees one output-channel group's weights
        const int xcd = bid & 7, within = bid >> 3;
        group = xcd % g.ngroups;
        t = within * g.xcd_per + xcd / g.ngroups;
    } else {
        group = bid % g.ngroups;
        t = bid / g.ngroups;
    }
    if (t >= g.tiles) return;
    const int tx = t % g.tiles_x; t /= g.tiles_x;
    const int ty = t % g.tiles_y; const int b = t / g.tiles_y;
    const int r0 = ty * kWavesT, c0 = tx * kTileWT, n0 = group * NB * 32;
    // (buffer-resource loads: a halo pixel outside the image -- the padding -- passes an offset beyond the tensor and reads zeros)
    const mdetr_rsrc xr = make_rsrc(x, static_cast<unsigned>(static_cast<int64_t>(d.B) * d.H * d.W * d.C * 2));
    const unsigned xb_off = static_cast<unsigned>(b * d.H * d.W) * static_cast<unsigned>(d.C * 2);

    for (int i = threadIdx.x; i < NB * 32; i += kWavesT * 64) shift_s[i] = (shift && n0 + i < d.N) ? shift[n0 + i] : 0.f;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

    constexpr int T = kWavesT * 64;
    // staging tasks: a 16-lane group takes 16 CONSECUTIVE LDS rows (weight rows, halo pixels in slot order) at one 16-byte piece --
    // rows are 20 dwords apart, so its ds_write_b128 lands on 16 distinct bank quads; the four groups of a wave take the four pieces
    // of the same rows, so one global load instruction still covers 64 contiguous bytes of each of its 16 rows.  (Consecutive
    // lanes = the pieces of one row put 4 rows on 16 lanes: rows 0 and 3 share banks; 27 - 38 % of the LDS cycles were conflicts,
    // profiles/r03_pmc_conv_strided_after.json.)
    constexpr int kGroup = 16 * kPiecesT;                                    // tasks per 16 rows
    constexpr int WROWS = (TS * NB * 32 + 15) / 16 * 16, HROWS = (HH * HWC + 15) / 16 * 16;
    constexpr int WP = (WROWS * kPiecesT + T - 1) / T;                       // weight pieces per thread
    constexpr int HP = (HROWS * kPiecesT + T - 1) / T;                       // halo pieces per thread
    bf16x8 wreg[WP];
    auto zero8 = []() { bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = static_cast<__bf16>(0.f);
        return v; };
    auto fetch_w = [&](int k0, int a) {                                      // [e][n][64 k] <- w[n0 + n][tap a][tap e][k0 .. k0 + 64)
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int p = threadIdx.x + j * T, piece = (p / 16) % kPiecesT, row = p / kGroup * 16 + p % 16;  // row = e * NB*32 + n
            const int e = row / (NB * 32), n = row - e * (NB * 32);
            bf16x8 v = zero8();
            if (row < TS * NB * 32 && n0 + n < d.N)
                v = *reinterpret_cast<const bf16x8 *>(w + static_cast<int64_t>(n0 + n) * d.w_sn + static_cast<int64_t>(d.ta0 + a * d.ta_step) * d.w_sa +
                                                      static_cast<int64_t>(d.te0 + e * d.te_step) * d.w_se + k0 + piece * 8);
            wreg[j] = v;
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int p = threadIdx.x + j * T, piece = (p / 16) % kPiecesT, row = p / kGroup * 16 + p % 16;
            if (row < TS * NB * 32) *reinterpret_cast<bf16x8 *>(wts + row * kPadT + piece * 8) = wreg[j];
        }
    };
    // halo: global -> registers -> LDS in one go at each slab boundary (HP pieces per thread: up to 19 with SI = 2, too many to
    // hold across a stage beside the accumulators); the weights of the next stage are what stays in flight during the products
    auto load_halo = [&](int k0) {
#pragma unroll
        for (int j0 = 0; j0 < HP; j0 += 8) {
            bf16x8 hreg[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = j0 + jj;
                if (j >= HP) break;
                const int p = threadIdx.x + j * T, piece = (p / 16) % kPiecesT, pix = p / kGroup * 16 + p % 16;   // pix = LDS row: hr * HWC + slot
                const int hr = pix / HWC, slot_ = pix - hr * HWC;
                const int hc = SI == 2 ? (slot_ < PLANE ? 2 * slot_ : 2 * (slot_ - PLANE) + 1) : slot_;
                const int r = SI * r0 + hr - d.PT, c = SI * c0 + hc - d.PL;
                // (a single tap at stride 2 reads only the even rows / columns of its halo: the others are not fetched)
                const bool used = !(SI == 2 && TS == 1 && (hc & 1)) && !(SI == 2 && TR == 1 && (hr & 1));
                const bool in = used && pix < HH * HWC && r >= 0 && r < d.H && c >= 0 && c < d.W;
                hreg[jj] = rsrc_load_bf16x8(xr, in ? static_cast<unsigned>(((r * d.W + c) * d.C + k0 + piece * 8) * 2) : kRsrcOob, xb_off);
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = j0 + jj;
                if (j >= HP) break;
                const int p = threadIdx.x + j * T, piece = (p / 16) % kPiecesT, pix = p / kGroup * 16 + p % 16;
                if (pix < HH * HWC) *reinterpret_cast<bf16x8 *>(halo + pix * kPadT + piece * 8) = hreg[jj];
            }
        }
    };

    const int slabs = d.C / kSlabT / g.ksplit, kbase = split * slabs * kSlabT;     // this workgroup's channel range
    const int stages = slabs * TR;
    fetch_w(kbase, 0);
    for (int q = 0; q < stages; ++q) {
        const int a = q % TR;
        __syncthreads();                                                     // the previous stage's LDS reads are done
        if (a == 0) load_halo(kbase + q / TR * kSlabT);
        store_w();
        if (q + 1 < stages) fetch_w(kbase + (q + 1) / TR * kSlabT, (q + 1) % TR);    // in flight during the products below
        __syncthreads();
#pragma unroll
        for (int e = 0; e < TS; ++e) {
            const int slot = SI == 2 ? (e & 1) * PLANE + col + (e >> 1) : col + e;       // halo column SI col + e of this lane's output pixel
            const __bf16 *hp = halo + ((SI * wave + a) * HWC + slot) * kPadT;
#pragma unroll
            for (int ks = 0; ks < kSlabT / 16; ++ks) {
                const bf16x8 xv = *reinterpret_cast<const bf16x8 *>(hp + ks * 16 + half * 8);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(wts + (e * NB * 32 + nb * 32 + col) * kPadT + ks * 16 + half * 8);
                    acc[nb] = mfma_bf16(wv, xv, acc[nb]);                    // Y^T[n][pixel]
                }
            }
        }
    }

    // ---- epilogue: lane = pixel; register quad q of block nb = channels 32 nb + 8 q + 4 half + 0..3 (conv3x3.hip)
    const int r = r0 + wave, c = c0 + col;
    if (g.part != nullptr) {                                                 // a split of the contraction: fp32 partial, no activation
        if (r < d.OH && c < d.OW) {
            float *pp = g.part + (((static_cast<int64_t>(split) * d.B + b) * d.OH + r) * d.OW + c) * d.N + n0 + 4 * half;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nn = nb * 32 + 8 * q + 4 * half;
                    if (n0 + nn < d.N) {
                        float o[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = acc[nb][4 * q + i] + (split == 0 ? shift_s[nn + i] : 0.f);
                        *reinterpret_cast<float4 *>(pp + nb * 32 + 8 * q) = make_float4(o[0], o[1], o[2], o[3]);
                    }
                }
        }
        return;
    }
    if (r < d.OH && c < d.OW) {
        __bf16 *yp = y + d.y_off + static_cast<int64_t>(b) * d.y_sb + static_cast<int64_t>(r) * d.y_sr + static_cast<int64_t>(c) * d.y_sc + n0 + 4 * half;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nn = nb * 32 + 8 * q + 4 * half;
                if (n0 + nn < d.N) {
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[nb][4 * q + i] + shift_s[nn + i];
                        if (RELU) v = v > 0.f ? v : 0.f;
                        o[i] = static_cast<__bf16>(v);
                    }
                    *reinterpret_cast<bf16x4 *>(yp + nb * 32 + 8 * q) = o;
                }
            }
    }
}

inline TapGeom geometry(const ConvTapsDims &d, int nb, int64_t &blocks)
{
    TapGeom g;
    g.d = d;
    g.tiles_x = (d.OW + kTileWT - 1) / kTileWT;
    g.tiles_y = (d.OH + kWavesT - 1) / kWavesT;
    g.tiles = d.B * g.tiles_x * g.tiles_y;
    g.ngroups = (d.N + nb * 32 - 1) / (nb * 32);
    g.xcd_per = (g.ngroups <= 8 && 8 % g.ngroups == 0) ? 8 / g.ngroups : 0;
    blocks = g.xcd_per ? 8ll * ((g.tiles + g.xcd_per - 1) / g.xcd_per) : static_cast<int64_t>(g.tiles) * g.ngroups;
    return g;
}

// output channels per workgroup = 32 NB: as many as the layer has (the halo is then staged once for all of them) -- unless that
// leaves most of the 256 CUs without a workgroup (the fourth pyramid level: 8 x 6 x 20 output pixels = 16 tiles)
inline int pick_nb(const ConvTapsDims &d, int nb_min)
{
    const int64_t tiles = static_cast<int64_t>(d.B) * ((d.OH + kWavesT - 1) / kWavesT) * ((d.OW + kTileWT - 1) / kTileWT);
    int nb = d.N >= 128 ? 4 : (d.N >= 64 ? 2 : 1);
    char tune_buf[8];
    if (const char *ev = tune_str("conv_taps_nb", tune_buf, sizeof(tune_buf))) {                     // tests / A-B runs: a fixed width (clamped to what exists)
        const int f = atoi(ev);
        if (f == 1 || f == 2 || f == 4) return f < nb_min ? nb_min : (f > nb ? nb : f);
    }
    while (nb > nb_min && tiles * ((d.N + nb * 32 - 1) / (nb * 32)) < 256) nb >>= 1;
    return nb < nb_min ? nb_min : nb;
}

template <int NB, bool RELU, int TR, int TS, int SI>
__global__ __launch_bounds__(kWavesT * 64)
void conv_taps_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w, const float *__restrict__ shift,
                      __bf16 *__restrict__ y, const TapGeom g)
{
    conv_taps_body<NB, RELU, TR, TS, SI>(x, w, shift, y, g, static_cast<int>(blockIdx.x));
}

// ---- the input gradient of a stride-2 convolution in ONE launch: the four pixel-parity classes side by side ----------------
// dX[2r + pi, 2c + pj] for (pi, pj) in {0, 1}^2; class (pi, pj) has (1 + pi) x (1 + pj) taps of a 3x3 kernel.  For a 1x1 kernel
// only the even pixels receive anything: the other three classes store zeros (every element of dX is written exactly once: no
// zero-fill pass).  Workgroups are numbered class by class, the four-tap class first (the longest workgroups start first).
struct Dgrad4Geom {
    TapGeom g[4];
    int blk0[5];                         // first workgroup of each class (in launch order), and the total
    int taps[4];                         // 0: store zeros, else (TR << 4) | TS
};

template <int NB>
__device__ __forceinline__ void zero_tile(__bf16 *__restrict__ y, const TapGeom &g, const int bid)
{
    const ConvTapsDims &d = g.d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    int group, t;
    if (g.xcd_per > 0) {
        const int xcd = bid & 7, within = bid >> 3;
        group = xcd % g.ngroups;
        t = within * g.xcd_per + xcd / g.ngroups;
    } else {
        group = bid % g.ngroups;
        t = bid / g.ngroups;
    }
    if (t >= g.tiles) return;
    const int tx = t % g.tiles_x; t /= g.tiles_x;
    const int ty = t % g.tiles_y; const int b = t / g.tiles_y;
    const int r = ty * kWavesT + wave, c = tx * kTileWT + col, n0 = group * NB * 32;
    if (r < d.OH && c < d.OW) {
        __bf16 *yp = y + d.y_off + static_cast<int64_t>(b) * d.y_sb + static_cast<int64_t>(r) * d.y_sr + static_cast<int64_t>(c) * d.y_sc + n0 + 4 * half;
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = static_cast<__bf16>(0.f);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (n0 + nb * 32 + 8 * q + 4 * half < d.N) *reinterpret_cast<bf16x4 *>(yp + nb * 32 + 8 * q) = o;
    }
}

template <int NB>
__global__ __launch_bounds__(kWavesT * 64)
void conv_dgrad4_kernel(const __bf16 *__restrict__ dy, const __bf16 *__restrict__ wt, __bf16 *__restrict__ dx, const Dgrad4Geom q)
{
    const int bid = static_cast<int>(blockIdx.x);
    const int cls = (bid >= q.blk0[1] ? 1 : 0) + (bid >= q.blk0[2] ? 1 : 0) + (bid >= q.blk0[3] ? 1 : 0);      // uniform
    const TapGeom &g = q.g[cls];
    const int local = bid - q.blk0[cls];
    switch (q.taps[cls]) {
    case 0x22: conv_taps_body<NB, false, 2, 2, 1>(dy, wt, nullptr, dx, g, local); break;
    case 0x21: conv_taps_body<NB, false, 2, 1, 1>(dy, wt, nullptr, dx, g, local); break;
    case 0x12: conv_taps_body<NB, false, 1, 2, 1>(dy, wt, nullptr, dx, g, local); break;
    case 0x11: conv_taps_body<NB, false, 1, 1, 1>(dy, wt, nullptr, dx, g, local); break;
    default: zero_tile<NB>(dx, g, local); break;
    }
}

template <int NB, bool RELU, int TR, int TS, int SI>
hipError_t launch(const void *x, const void *w, const float *shift, void *y, const ConvTapsDims &d, hipStream_t st)
{
    constexpr int HH = SI * (kWavesT - 1) + TR, HWC = SI * (kTileWT - 1) + TS;
    constexpr size_t lds = static_cast<size_t>(HH) * HWC * kPadT * 2 + static_cast<size_t>(TS) * NB * 32 * kPadT * 2 + NB * 32 * 4;
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    auto kern = conv_taps_kernel<NB, RELU, TR, TS, SI>;
    static bool attr_set[64] = {};                           // per device
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = -1;
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    int64_t blocks = 0;
    const TapGeom g = geometry(d, NB, blocks);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(kWavesT * 64), lds, st, static_cast<const __bf16 *>(x),
                       static_cast<const __bf16 *>(w), shift, static_cast<__bf16 *>(y), g);
    return hipGetLastError();
}

// the split form (see TapGeom): 32 output channels per workgroup, ksplit workgroups per (tile, channel group)
template <int TR, int TS, int SI>
hipError_t launch_split(const void *x, const void *w, const float *shift, float *part, const ConvTapsDims &d, int ksplit, hipStream_t st)
{
    constexpr int NB = 1;
    constexpr int HH = SI * (kWavesT - 1) + TR, HWC = SI * (kTileWT - 1) + TS;
    constexpr size_t lds = static_cast<size_t>(HH) * HWC * kPadT * 2 + static_cast<size_t>(TS) * NB * 32 * kPadT * 2 + NB * 32 * 4;
    auto kern = conv_taps_kernel<NB, false, TR, TS, SI>;
    static bool attr_set[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = -1;
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    int64_t blocks = 0;
    TapGeom g = geometry(d, NB, blocks);
    g.ksplit = ksplit;
    g.inner_blocks = static_cast<int>(blocks);
    g.part = part;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks * ksplit)), dim3(kWavesT * 64), lds, st, static_cast<const __bf16 *>(x),
                       static_cast<const __bf16 *>(w), shift, static_cast<__bf16 *>(nullptr), g);
    return hipGetLastError();
}

template <int TR, int TS, int SI>
hipError_t by_width(const void *x, const void *w, const float *shift, void *y, const ConvTapsDims &d, bool relu, hipStream_t st)
{
    const int nb = pick_nb(d, SI == 2 && TR == 3 ? 1 : 2);
    if (nb == 4) return relu ? launch<4, true, TR, TS, SI>(x, w, shift, y, d, st) : launch<4, false, TR, TS, SI>(x, w, shift, y, d, st);
    if constexpr (SI == 2 && TR == 3) {
        if (nb == 1) return relu ? launch<1, true, TR, TS, SI>(x, w, shift, y, d, st) : launch<1, false, TR, TS, SI>(x, w, shift, y, d, st);
    }
    return relu ? launch<2, true, TR, TS, SI>(x, w, shift, y, d, st) : launch<2, false, TR, TS, SI>(x, w, shift, y, d, st);
}

template <int NB>
hipError_t launch_dgrad4(const void *dy, const void *wt, void *dx, const Dgrad4Geom &q, hipStream_t st)
{
    constexpr int HH = kWavesT - 1 + 2, HWC = kTileWT - 1 + 2;                 // the 2x2 class needs the most
    constexpr size_t lds = static_cast<size_t>(HH) * HWC * kPadT * 2 + static_cast<size_t>(2) * NB * 32 * kPadT * 2 + NB * 32 * 4;
    auto kern = conv_dgrad4_kernel<NB>;
    static bool attr_set[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = -1;
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(q.blk0[4])), dim3(kWavesT * 64), lds, st, static_cast<const __bf16 *>(dy),
                       static_cast<const __bf16 *>(wt), static_cast<__bf16 *>(dx), q);
    return hipGetLastError();
}

}  // namespace

bool conv_taps_supported(const ConvTapsDims &d, const void *x, const void *w, const void *y)
{
    const auto al = [](const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    const bool shape = (d.SI == 2 && ((d.TR == 3 && d.TS == 3) || (d.TR == 1 && d.TS == 1))) ||
                       (d.SI == 1 && d.TR >= 1 && d.TR <= 2 && d.TS >= 1 && d.TS <= 2);
    return shape && d.B > 0 && d.H > 0 && d.W > 0 && d.OH > 0 && d.OW > 0 && d.C > 0 && d.C % 64 == 0 && d.N > 0 && d.N % 32 == 0 &&
           al(x, 16) && al(w, 16) && al(y, 8) && d.w_sn % 8 == 0 && d.w_sa % 8 == 0 && d.w_se % 8 == 0 &&
           d.y_off % 4 == 0 && d.y_sb % 4 == 0 && d.y_sr % 4 == 0 && d.y_sc % 4 == 0 && d.PT >= 0 && d.PL >= 0 &&
           static_cast<int64_t>(d.B) * d.H * d.W * d.C < (1ll << 30) &&
           static_cast<int64_t>(d.B) * ((d.OH + 3) / 4) * ((d.OW + 31) / 32) * ((d.N + 31) / 32) < (1ll << 30);
}

bool conv_dgrad_s2_supported(int B, int OH, int OW, int N, int H, int W, int C, int K, const void *dy, const void *wt, const void *dx)
{
    const auto al = [](const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    return B > 0 && OH > 0 && OW > 0 && H > 0 && W > 0 && (K == 3 || K == 1) && N > 0 && N % 64 == 0 && C > 0 && C % 32 == 0 &&
           OH == (H - 1) / 2 + 1 && OW == (W - 1) / 2 + 1 && al(dy, 16) && al(wt, 16) && al(dx, 8) &&
           static_cast<int64_t>(B) * OH * OW * N < (1ll << 30) && static_cast<int64_t>(B) * H * W * C < (1ll << 31);
}

// dy [B, OH, OW, N], wt [C, K, K, N] (the weight with its channel axes swapped; taps NOT mirrored), dx [B, H, W, C]
hipError_t conv_dgrad_s2_launch(const void *dy, const void *wt, void *dx, int B, int OH, int OW, int N, int H, int W, int C, int K, hipStream_t st)
{
    Dgrad4Geom q;
    ConvTapsDims probe;
    probe.B = B; probe.OH = (H + 1) / 2; probe.OW = (W + 1) / 2; probe.N = C;
    const int nb = pick_nb(probe, 2);
    const int order[4] = {3, 2, 1, 0};                                        // class = 2 pi + pj; the four-tap class first
    int blk = 0;
    for (int k = 0; k < 4; ++k) {
        const int cls = order[k], pi = cls >> 1, pj = cls & 1;
        ConvTapsDims d;
        d.B = B; d.H = OH; d.W = OW; d.C = N;
        d.OH = (H - pi + 1) / 2; d.OW = (W - pj + 1) / 2; d.N = C;
        d.SI = 1; d.PT = 0; d.PL = 0;
        d.y_off = (static_cast<int64_t>(pi) * W + pj) * C; d.y_sb = static_cast<int64_t>(H) * W * C; d.y_sr = 2ll * W * C; d.y_sc = 2ll * C;
        d.w_sn = static_cast<int64_t>(K) * K * N; d.w_sa = static_cast<int64_t>(K) * N; d.w_se = N;
        if (K == 3) {
            // even coordinate 2r: the centre tap on dY[r]; odd coordinate 2r + 1: tap 2 on dY[r], then tap 0 on dY[r + 1]
            d.TR = 1 + pi; d.TS = 1 + pj;
            d.ta0 = pi ? 2 : 1; d.ta_step = pi ? -2 : 0;
            d.te0 = pj ? 2 : 1; d.te_step = pj ? -2 : 0;
            q.taps[k] = (d.TR << 4) | d.TS;
        } else {
            d.TR = d.TS = 1; d.ta0 = d.te0 = 0; d.ta_step = d.te_step = 0;
            q.taps[k] = cls == 0 ? 0x11 : 0;                                  // a 1x1 / stride-2 convolution reads the even pixels only
        }
        int64_t blocks = 0;
        q.g[k] = geometry(d, nb, blocks);
        if (d.OH <= 0 || d.OW <= 0) blocks = 0;
        q.blk0[k] = blk;
        blk += static_cast<int>(blocks);
    }
    q.blk0[4] = blk;
    if (blk == 0) return hipSuccess;
    // every output pixel of dY meets every (tap, channel pair) once: the four parity classes together are the dense product
    ProfileScope prof(9, conv_mflop(static_cast<int64_t>(B) * OH * OW, static_cast<int64_t>(C) * N * K * K), st, 2.0 * B * OH * OW * C * N * K * K / 1e6,
                      (2.0 * B * (static_cast<double>(OH) * OW * N + 4.0 * OH * OW * C) + 2.0 * K * K * N * C) / 1e3);
    return nb == 4 ? launch_dgrad4<4>(dy, wt, dx, q, st) : launch_dgrad4<2>(dy, wt, dx, q, st);
}

bool conv_taps_split_supported(const ConvTapsDims &d, int ksplit)
{
    // the 3x3 / stride-2 forward form only (the one the fourth pyramid level needs); whole slabs per split
    return d.SI == 2 && d.TR == 3 && d.TS == 3 && ksplit >= 2 && ksplit <= 64 && d.C % (kSlabT * ksplit) == 0 && d.N % 4 == 0;
}

hipError_t conv_taps_split_launch(const void *x, const void *w, const float *shift, float *part, const ConvTapsDims &d, int ksplit, hipStream_t st)
{
    ProfileScope prof(9, conv_mflop(static_cast<int64_t>(d.B) * d.OH * d.OW, static_cast<int64_t>(d.C) * d.N * d.TR * d.TS), st,
                      2.0 * d.B * d.OH * d.OW * d.C * d.N * d.TR * d.TS / 1e6,
                      (2.0 * d.B * (static_cast<double>(d.H) * d.W * d.C) + 4.0 * ksplit * d.B * d.OH * d.OW * d.N + 2.0 * d.TR * d.TS * d.N * d.C) / 1e3);
    return launch_split<3, 3, 2>(x, w, shift, part, d, ksplit, st);
}

hipError_t conv_taps_launch(const void *x, const void *w, const float *shift, void *y, const ConvTapsDims &d, bool relu, hipStream_t st)
{
    ProfileScope prof(9, conv_mflop(static_cast<int64_t>(d.B) * d.OH * d.OW, static_cast<int64_t>(d.C) * d.N * d.TR * d.TS), st,
                      2.0 * d.B * d.OH * d.OW * d.C * d.N * d.TR * d.TS / 1e6,
                      (2.0 * d.B * (static_cast<double>(d.H) * d.W * d.C + static_cast<double>(d.OH) * d.OW * d.N) + 2.0 * d.TR * d.TS * d.N * d.C) / 1e3);
    if (d.SI == 2) return d.TR == 3 ? by_width<3, 3, 2>(x, w, shift, y, d, relu, st) : by_width<1, 1, 2>(x, w, shift, y, d, relu, st);
    if (d.TR == 1) return d.TS == 1 ? by_width<1, 1, 1>(x, w, shift, y, d, relu, st) : by_width<1, 2, 1>(x, w, shift, y, d, relu, st);
    return d.TS == 1 ? by_width<2, 1, 1>(x, w, shift, y, d, relu, st) : by_width<2, 2, 1>(x, w, shift, y, d, relu, st);
}

}  // namespace mdetr
