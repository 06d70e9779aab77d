// monodetr_amd/csrc/conv3x3.hip -- 3x3 / stride 1 / pad 1 convolution of a channels-last bf16 activation as an implicit
// GEMM on the matrix cores, im2col done in LDS, with the frozen-BN shift and the ReLU in the epilogue.
//
// The ResNet-50 body has 13 such convolutions (torchvision Bottleneck.conv2 behind lib/models/monodetr/backbone.py:100-102;
// 18.1 GFLOP each at B = 8, whatever the stage), forward and -- with the taps mirrored and the channel axes swapped --
// input gradient.  MIOpen's implicit-GEMM kernels take 64-70 us for one of them in profiles/r01h (~ 270 TFLOP/s, 11 % of
// the dense bf16 rate) and leave shift, ReLU and casts to separate passes.
//
//   y[b, r, c, n] = act( shift[n] + sum_{t, s, k} x[b, r + t - 1, c + s - 1, k] * w[n, t, s, k] )
//   x [B, H, W, C] (C % 64 == 0), w [N, 3, 3, C] (the channels_last layout of an [N, C, 3, 3] weight), y [B, H, W, N]
//
// A workgroup (4 waves) owns 4 output rows x 32 columns x NB*32 output channels of one image; wave i owns row i, lane & 31 a
// column.  Per 64-channel slab of the input the (4 + 2) x (32 + 2) pixel halo is staged in LDS once -- zero outside the
// image: this IS the padding -- and serves all nine taps as shifted reads: tap (t, s) of output pixel (i, j) is halo pixel
// (i + t, j + s).  The weights of one tap row (3 taps x NB*32 channels x 64 k) follow through LDS.  Products are issued
// transposed, Y^T[n][pixel] = W[n][:] . X[pixel][:], with v_mfma_f32_32x32x16_bf16 (fragment conventions of token_gemm.hip /
// attn.hip, validated there): a lane's accumulator quad holds four consecutive output channels of ITS pixel, so shift, ReLU and
// the bf16 rounding happen in registers and leave as 8-byte stores.  LDS rows are padded to 72 bf16 (36 dwords: the 16 rows
// of a ds_read_b128 lane group fall on distinct bank quads).
// The next stage's global loads are staged in registers while the current stage's products run (one LDS buffer).
// Algorithmic bytes = 2 B H W (C + N) + 18 N C;  flops = 18 B H W C N.  MFMA-bound by design (LDS-read-bound in this first
// version: 5 ds_read_b128 per 4 MFMAs).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mdetr_wave.h>

#include "conv3x3.h"
#include "msda.h"       // profile scopes

namespace mdetr {
namespace {

constexpr int kWavesC = 4;               // = output rows per workgroup
constexpr int kTileW = 32;               // output columns per workgroup (one per lane & 31)
constexpr int kSlab = 64;                // input channels per LDS slab
constexpr int kPad = kSlab + 8;          // 72 bf16 per LDS row
constexpr int kHaloH = kWavesC + 2, kHaloW = kTileW + 2;

struct ConvDims {
    int B, H, W, C, N;
    int tiles_x, tiles_y;                // column / row tiles per image
    int tiles;                           // B * tiles_x * tiles_y
    int ngroups;                         // output-channel groups of NB*32
    int xcd_per;                         // 8 / ngroups when that is whole (XCD-aware numbering below), else 0
    int mirror;                          // taps read mirrored: w[n][2 - t][2 - s][k] (the input gradient: no mirrored weight copy)
};

template <int NB, bool RELU>
__global__ __launch_bounds__(kWavesC * 64)
void conv3x3_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w, const float *__restrict__ shift,
                    __bf16 *__restrict__ y, const ConvDims d)
{
    MDETR_DYNAMIC_LDS(unsigned char, conv_smem);
    __bf16 *halo = reinterpret_cast<__bf16 *>(conv_smem);                    // [kHaloH][kHaloW][kPad]
    __bf16 *wts = halo + kHaloH * kHaloW * kPad;                            // [3 taps][NB*32][kPad]
    float *shift_s = reinterpret_cast<float *>(wts + 3 * NB * 32 * kPad);    // [NB*32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    // Workgroup id -> (output-channel group, pixel tile).  The dispatcher deals consecutive workgroups round-robin to the 8
    // XCDs, each with its own L2: with ngroups in {1, 2, 4, 8} the group is made a function of id % 8, so an XCD only ever
    // touches the weights of 8 / ngroups ... of ONE group (1.2 MB of the 4.7 MB at 512 channels) instead of all of them.
    int group, t;
    if (d.xcd_per > 0) {
        const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
        group = xcd % d.ngroups;
        t = within * d.xcd_per + xcd / d.ngroups;
    } else {
        group = blockIdx.x % d.ngroups;
        t = blockIdx.x / d.ngroups;
    }
    if (t >= d.tiles) return;                                               // padding of the XCD numbering (whole workgroup, before any barrier)
    const int tx = t % d.tiles_x; t /= d.tiles_x;
    const int ty = t % d.tiles_y; const int b = t / d.tiles_y;
    const int r0 = ty * kWavesC, c0 = tx * kTileW, n0 = group * NB * 32;
    const __bf16 *xb = x + static_cast<int64_t>(b) * d.H * d.W * d.C;

    for (int i = threadIdx.x; i < NB * 32; i += kWavesC * 64) shift_s[i] = (shift && n0 + i < d.N) ? shift[n0 + i] : 0.f;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

    // Staging registers: the NEXT stage's weights (and, at a slab boundary, halo) are requested from global memory before
    // the current stage's products are issued and written to LDS after them, so a whole stage of loads is in flight during
    // the matrix work (one LDS buffer, two barriers per stage).  Stage q = (slab q / 3, tap row q % 3).
    constexpr int WP = 3 * NB * 32 * (kSlab / 8) / (kWavesC * 64);          // weight pieces per thread (3 NB)
    constexpr int HP = (kHaloH * kHaloW * (kSlab / 8) + kWavesC * 64 - 1) / (kWavesC * 64);     // halo pieces per thread (7)
    bf16x8 wreg[WP], hreg[HP];
    auto fetch_w = [&](int k0, int tr) {                                    // [s][n][64 k] <- w[n0 + n][tr][s][k0 .. k0 + 64)
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int p = threadIdx.x + j * kWavesC * 64, piece = p & 7, row = p >> 3;      // row = s * NB*32 + n
            const int s = row / (NB * 32), n = row - s * (NB * 32);
            bf16x8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = static_cast<__bf16>(0.f);
            if (n0 + n < d.N)
                v = *reinterpret_cast<const bf16x8 *>(w + ((static_cast<int64_t>(n0 + n) * 3 + (d.mirror ? 2 - tr : tr)) * 3 + (d.mirror ? 2 - s : s)) * d.C + k0 + piece * 8);
            wreg[j] = v;
        }
    };
    auto fetch_h = [&](int k0) {                                            // (4 + 2) x (32 + 2) pixels x 64 channels, zero outside the image
#pragma unroll
        for (int j = 0; j < HP; ++j) {
            const int p = threadIdx.x + j * kWavesC * 64, piece = p & 7, pix = p >> 3;
            const int hr = pix / kHaloW, hc = pix - hr * kHaloW;
            const int r = r0 + hr - 1, c = c0 + hc - 1;
            bf16x8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = static_cast<__bf16>(0.f);
            if (pix < kHaloH * kHaloW && r >= 0 && r < d.H && c >= 0 && c < d.W)
                v = *reinterpret_cast<const bf16x8 *>(xb + (static_cast<int64_t>(r) * d.W + c) * d.C + k0 + piece * 8);
            hreg[j] = v;
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int p = threadIdx.x + j * kWavesC * 64;
            *reinterpret_cast<bf16x8 *>(wts + (p >> 3) * kPad + (p & 7) * 8) = wreg[j];
        }
    };
    auto store_h = [&]() {
#pragma unroll
        for (int j = 0; j < HP; ++j) {
            const int p = threadIdx.x + j * kWavesC * 64;
            if (p < kHaloH * kHaloW * (kSlab / 8)) *reinterpret_cast<bf16x8 *>(halo + (p >> 3) * kPad + (p & 7) * 8) = hreg[j];
        }
    };

    const int stages = d.C / kSlab * 3;
    fetch_h(0);
    fetch_w(0, 0);
    for (int q = 0; q < stages; ++q) {
        const int tr = q % 3;
        __syncthreads();                                                    // the previous stage's LDS reads are done
        if (tr == 0) store_h();
        store_w();
        if (q + 1 < stages) {                                               // in flight during the products below
            const int nq = q + 1, ntr = nq % 3, nk0 = nq / 3 * kSlab;
            if (ntr == 0) fetch_h(nk0);
            fetch_w(nk0, ntr);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const __bf16 *hp = halo + ((wave + tr) * kHaloW + col + s) * kPad;          // input pixel of tap (tr, s) for this lane's output pixel
#pragma unroll
            for (int ks = 0; ks < kSlab / 16; ++ks) {
                const bf16x8 xv = *reinterpret_cast<const bf16x8 *>(hp + ks * 16 + half * 8);           // B operand: this lane's pixel, 8 k
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(wts + (s * NB * 32 + nb * 32 + col) * kPad + ks * 16 + half * 8);
                    acc[nb] = mfma_bf16(wv, xv, acc[nb]);                   // Y^T[n][pixel]
                }
            }
        }
    }

    // ---- epilogue: lane = pixel (row wave, column lane & 31); register quad g of block nb = channels 32 nb + 8 g + 4 half + 0..3
    const int r = r0 + wave, c = c0 + col;
    if (r < d.H && c < d.W) {
        __bf16 *yp = y + ((static_cast<int64_t>(b) * d.H + r) * d.W + c) * d.N + n0 + 4 * half;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nn = nb * 32 + 8 * g + 4 * half;
                if (n0 + nn < d.N) {                                        // N % 32 == 0: a quad is in or out as a whole
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[nb][4 * g + i] + shift_s[nn + i];
                        if (RELU) v = v > 0.f ? v : 0.f;
                        o[i] = static_cast<__bf16>(v);
                    }
                    *reinterpret_cast<bf16x4 *>(yp + nb * 32 + 8 * g) = o;
                }
            }
    }
}

template <int NB, bool RELU>
hipError_t launch(const void *x, const void *w, const float *shift, void *y, const ConvDims &d, hipStream_t st)
{
    constexpr size_t lds = static_cast<size_t>(kHaloH) * kHaloW * kPad * 2 + 3 * NB * 32 * kPad * 2 + NB * 32 * 4;
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    auto kern = conv3x3_kernel<NB, RELU>;
    static bool attr_set[64] = {};                           // the attribute is per device: one process may drive several GPUs
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = -1;
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    ConvDims g = d;
    g.tiles = d.B * d.tiles_x * d.tiles_y;
    g.ngroups = (d.N + NB * 32 - 1) / (NB * 32);
    g.xcd_per = (g.ngroups <= 8 && 8 % g.ngroups == 0) ? 8 / g.ngroups : 0;
    const int64_t blocks = g.xcd_per ? 8ll * ((g.tiles + g.xcd_per - 1) / g.xcd_per) : static_cast<int64_t>(g.tiles) * g.ngroups;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(kWavesC * 64), lds, st, static_cast<const __bf16 *>(x),
                       static_cast<const __bf16 *>(w), shift, static_cast<__bf16 *>(y), g);
    return hipGetLastError();
}

}  // namespace

bool conv3x3_supported(int B, int H, int W, int C, int N, const void *x, const void *w, const void *y)
{
    const auto al = [](const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    return B > 0 && H > 0 && W > 0 && C > 0 && C % 64 == 0 && N > 0 && N % 32 == 0 && al(x, 16) && al(w, 16) && al(y, 8) &&
           static_cast<int64_t>(B) * ((H + 3) / 4) * ((W + 31) / 32) * ((N + 31) / 32) < (1ll << 30);
}

hipError_t conv3x3_launch(const void *x, const void *w, const float *shift, void *y, int B, int H, int W, int C, int N, bool relu,
                          hipStream_t st, bool mirror)
{
    ConvDims d{B, H, W, C, N, (W + kTileW - 1) / kTileW, (H + kWavesC - 1) / kWavesC, 0, 0, 0, mirror ? 1 : 0};
    ProfileScope prof(9, conv_mflop(static_cast<int64_t>(B) * H * W, static_cast<int64_t>(C) * N * 9), st, 18.0 * B * H * W * C * N / 1e6,
                      (2.0 * B * H * W * (C + N) + 18.0 * N * C) / 1e3);
    // 128 output channels per workgroup where the layer has them (the halo is then read once per 128 channels); 64 for the
    // 64-channel stage
    int nb = N >= 128 ? 4 : (N >= 64 ? 2 : 1);
    // 85 KB of LDS at 128 channels = ONE workgroup (4 waves) per CU; 57 KB at 64 (two per CU), 43 KB at 32 (three).  A problem
    // with few pixel tiles (layer3: 144, layer4: 48 at B = 8) is better served by narrow channel blocks -- more workgroups, more
    // waves per SIMD to overlap the LDS reads with -- than by reading the halo once per 128 channels: measured at B = 8
    // (profiles/r03o_conv3x3_nb*.json) layer3 61 / 42 / 36 us and layer4 58 / 47 / 41 us at 128 / 64 / 32 channels per
    // workgroup, layer2 (480 tiles) 38 / 32 / 33 us.  Narrow until there are ~900 workgroups.
    const int64_t tiles = static_cast<int64_t>(B) * d.tiles_x * d.tiles_y;
    while (nb > 1 && tiles * ((N + nb * 32 - 1) / (nb * 32)) < 900) nb >>= 1;
    if (nb == 4) return relu ? launch<4, true>(x, w, shift, y, d, st) : launch<4, false>(x, w, shift, y, d, st);
    if (nb == 2) return relu ? launch<2, true>(x, w, shift, y, d, st) : launch<2, false>(x, w, shift, y, d, st);
    return relu ? launch<1, true>(x, w, shift, y, d, st) : launch<1, false>(x, w, shift, y, d, st);
}

}  // namespace mdetr
