// monodetr_amd/csrc/conv_stem.hip -- the ResNet stem: 7x7 / stride 2 / pad 3 convolution of the 3-channel image into 64 channels,
// frozen-BN shift + ReLU in the epilogue, as an implicit GEMM on the matrix cores (torchvision ResNet.conv1 -> bn1 -> relu behind
// lib/models/monodetr/backbone.py:93-106; frozen, so forward only).  MIOpen runs it in 116 us at B = 8, 384 x 1280
// (profiles/r02v: igemm_fwd_gtcx35 ... 256x64x8), 18.5 GFLOP: 160 TFLOP/s.
//
//   y[b, r, c, n] = relu( shift[n] + sum_{t < 7, e < 7, ch < 3} x[b, 2r + t - 3, 2c + e - 3, ch] * w[n, t, e, ch] )
//
// With 3 channels a pixel is 6 bytes: there is no 64-channel slab to stage.  The contraction index is laid out per tap ROW: the
// 7 taps x 3 channels of one input row are 21 CONSECUTIVE bf16 of the channels-last image, padded to 24 = three 8-element
// operand groups, so K = 7 x 24 = 168 (+ 8 zeros = 11 k-steps of 16).  The input window of a tile -- 13 rows x 70 pixels -- is
// copied into LDS as it lies in memory (2-byte buffer loads: zero outside the image = the padding); a lane's B operand for
// group (t, m) is the 8 bf16 at element 6 c + 8 m of window row 2 r + t: four aligned 4-byte LDS reads (12 c + 16 m bytes; the
// lanes of a half-wave are 3 banks apart: conflict-free).  The pad elements 21..23 of a group are the NEXT pixel's channels
// (finite data) against zero weights.  The packed weight [64][7][24] (+ 8 zeros) is built once by the caller -- the stem is
// frozen -- and held in LDS by the workgroup across its column tiles.
// Workgroup = 4 waves = 4 output rows x 32 output columns x 64 channels per tile, looping over kTilesPerWg column tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mdetr_wave.h>

#include "conv_stem.h"
#include "msda.h"       // profile scopes

namespace mdetr {
namespace {

constexpr int kWavesS = 4, kTileWS = 32;
constexpr int kKS = 176;                  // packed contraction length: 7 x 24 + 8 zeros
constexpr int kWRow = kKS + 8;            // bf16 per weight row in LDS (23 sixteen-byte slots: odd)
constexpr int kWinRows = 2 * (kWavesS - 1) + 7;          // 13
constexpr int kWinPix = 2 * (kTileWS - 1) + 8;           // 70 pixels: 69 needed + the one the pad elements of the last lane fall on
constexpr int kWinEl = kWinPix * 3;                      // 210
constexpr int kWinRow = 216;                             // bf16 per window row in LDS
constexpr int kTilesPerWg = 5;

struct StemDims {
    int B, H, W, OH, OW, tiles_x, groups_x, tiles_y;
};

__global__ __launch_bounds__(kWavesS * 64)
void conv_stem_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ wp, const float *__restrict__ shift,
                      __bf16 *__restrict__ y, const StemDims d)
{
    __shared__ __attribute__((aligned(16))) __bf16 wts[64 * kWRow];
    __shared__ __attribute__((aligned(16))) unsigned short win[kWinRows * kWinRow];
    __shared__ float shift_s[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
    int id = blockIdx.x;
    const int gx = id % d.groups_x; id /= d.groups_x;
    const int ty = id % d.tiles_y; const int b = id / d.tiles_y;
    const int r0 = ty * kWavesS;

    for (int i = threadIdx.x; i < 64 * (kKS / 8); i += kWavesS * 64) {       // packed weight -> LDS, 16 bytes at a time
        const int n = i / (kKS / 8), piece = i - n * (kKS / 8);
        *reinterpret_cast<bf16x8 *>(wts + n * kWRow + piece * 8) = *reinterpret_cast<const bf16x8 *>(wp + n * kKS + piece * 8);
    }
    if (threadIdx.x < 64) shift_s[threadIdx.x] = shift ? shift[threadIdx.x] : 0.f;

    const mdetr_rsrc xr = make_rsrc(x, static_cast<unsigned>(static_cast<int64_t>(d.B) * d.H * d.W * 6));
    // The window of the NEXT column tile is requested while this tile's products run and stored behind the barrier: one 2-byte load
    // per thread and loop iteration straight into LDS -- the first version -- was 11 dependent round trips per tile ahead of the
    // first matrix instruction.  (A tile beyond the row reads nothing: offsets beyond the resource.)
    constexpr int kWinLoads = (kWinRows * kWinEl + kWavesS * 64 - 1) / (kWavesS * 64);
    unsigned short wreg[kWinLoads];
    auto fetch_window = [&](int tx) {
        const int c0n = tx * kTileWS;
#pragma unroll
        for (int j = 0; j < kWinLoads; ++j) {
            const int i = threadIdx.x + j * kWavesS * 64;
            const int wr = i / kWinEl, el = i - wr * kWinEl;
            const int row = 2 * r0 + wr - 3, gel = (2 * c0n - 3) * 3 + el;   // element of the image row
            const bool in = tx < d.tiles_x && i < kWinRows * kWinEl && row >= 0 && row < d.H && gel >= 0 && gel < d.W * 3;
            wreg[j] = rsrc_load_u16(xr, in ? static_cast<unsigned>((row * d.W * 3 + gel) * 2) : kRsrcOob,
                                    static_cast<unsigned>(b) * static_cast<unsigned>(d.H * d.W * 6));
        }
    };
    fetch_window(gx * kTilesPerWg);
    for (int tt = 0; tt < kTilesPerWg; ++tt) {
        const int tx = gx * kTilesPerWg + tt;
        if (tx >= d.tiles_x) break;                                          // uniform
        const int c0 = tx * kTileWS;
        __syncthreads();                                                     // the previous tile's window reads are done (and, first time, nothing)
        // window: rows 2 r0 - 3 .. + 12, elements (2 c0 - 3) * 3 .. + 209 of each
#pragma unroll
        for (int j = 0; j < kWinLoads; ++j) {
            const int i = threadIdx.x + j * kWavesS * 64;
            if (i < kWinRows * kWinEl) win[(i / kWinEl) * kWinRow + (i % kWinEl)] = wreg[j];
        }
        if (tt + 1 < kTilesPerWg) fetch_window(tx + 1);                      // in flight during the products below
        __syncthreads();

        f32x16 acc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < kKS / 16; ++ks) {
            int gidx = 2 * ks + half;                                        // operand group (t, m) = (g / 3, g % 3)
            if (gidx > 20) gidx = 20;                                        // the 8 zero weights at the end: any finite data
            const int t = gidx / 3, m = gidx - 3 * t;
            const unsigned *src = reinterpret_cast<const unsigned *>(win + (2 * wave + t) * kWinRow + 6 * col + 8 * m);
            unsigned q[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = src[i];
            bf16x8 xv;
            __builtin_memcpy(&xv, q, 16);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(wts + (nb * 32 + col) * kWRow + ks * 16 + half * 8);
                acc[nb] = mfma_bf16(wv, xv, acc[nb]);                        // Y^T[n][pixel]
            }
        }
        // epilogue: lane = pixel (row wave, column lane & 31); register quad q of block nb = channels 32 nb + 8 q + 4 half + 0..3
        const int r = r0 + wave, c = c0 + col;
        if (r < d.OH && c < d.OW) {
            __bf16 *yp = y + ((static_cast<int64_t>(b) * d.OH + r) * d.OW + c) * 64 + 4 * half;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[nb][4 * qd + i] + shift_s[nb * 32 + 8 * qd + 4 * half + i];
                        o[i] = static_cast<__bf16>(v > 0.f ? v : 0.f);
                    }
                    *reinterpret_cast<bf16x4 *>(yp + nb * 32 + 8 * qd) = o;
                }
        }
    }
}

}  // namespace

bool conv_stem_supported(int B, int H, int W, const void *x, const void *wp, const void *y)
{
    const auto al = [](const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    return B > 0 && H > 0 && W > 0 && al(x, 2) && al(wp, 16) && al(y, 8) && static_cast<int64_t>(B) * H * W * 6 < (1ll << 31);
}

hipError_t conv_stem_launch(const void *x, const void *wp, const float *shift, void *y, int B, int H, int W, hipStream_t st)
{
    StemDims d;
    d.B = B; d.H = H; d.W = W;
    d.OH = (H - 1) / 2 + 1; d.OW = (W - 1) / 2 + 1;
    d.tiles_x = (d.OW + kTileWS - 1) / kTileWS;
    d.groups_x = (d.tiles_x + kTilesPerWg - 1) / kTilesPerWg;
    d.tiles_y = (d.OH + kWavesS - 1) / kWavesS;
    const int64_t blocks = static_cast<int64_t>(B) * d.tiles_y * d.groups_x;
    ProfileScope prof(9, conv_mflop(static_cast<int64_t>(B) * d.OH * d.OW, 64 * 3 * 49), st, 2.0 * B * d.OH * d.OW * 64 * 147 / 1e6,
                      (2.0 * B * (static_cast<double>(d.H) * d.W * 3 + static_cast<double>(d.OH) * d.OW * 64)) / 1e3);
    hipLaunchKernelGGL(conv_stem_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kWavesS * 64), 0, st, static_cast<const __bf16 *>(x),
                       static_cast<const __bf16 *>(wp), shift, static_cast<__bf16 *>(y), d);
    return hipGetLastError();
}

}  // namespace mdetr
