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
// A workgroup (4 waves) owns 4 output rows x 32 columns x NB*32 output channels of one image.  Per 64-channel slab of the input
// the (4 + 2) x (32 + 2) pixel halo is staged in LDS once -- zero outside the image: this IS the padding -- and serves all nine
// taps as shifted reads: tap (t, s) of output pixel (i, j) is halo pixel (i + t, j + s).  The weights of one tap row (3 taps x
// NB*32 channels x 64 k) follow through LDS.  Products are issued transposed, Y^T[n][pixel] = W[n][:] . X[pixel][:], with
// v_mfma_f32_32x32x16_bf16 (fragment conventions of token_gemm.hip / attn.hip, validated there): a lane's accumulator quad holds
// four consecutive output channels of ITS pixel, so shift, ReLU and the bf16 rounding happen in registers and leave as 8-byte
// stores.  LDS rows are padded to 72 bf16 (36 dwords: the 16 rows of a ds_read_b128 lane group fall on distinct bank quads).
//
// The 32 pixels of a wave are a WR x WC block, WC in {32, 16, 8}; the four blocks of a workgroup lie GC side by side and 4 / GC one
// below the other (templates): tiles of 4 x 32 or 8 x 16 pixels.  A wave whose block lies outside the image takes no part in the
// products (it still stages its share of the tile).  Two things are chosen with the shape (`choose` below):
//   * how many matrix instructions work on empty columns: at W = 80 (layer3) 32-wide blocks spend 18 blocks on 15 blocks of
//     pixels, at W = 40 (layer4) 8 on 5;
//   * how many ROUNDS of workgroups the launch takes.  A workgroup's duration hardly depends on its neighbours on the CU, so a
//     launch lasts rounds x duration: layer3 in 4 x 32 tiles at 64 channels is 576 workgroups for 512 places -- 13 us with the
//     device full, then 17 us for the last 64 workgroups --, in 8 x 16 tiles 480: one round (30.2 -> 23.3 us, profiles/r06y_*).
// The halo's LDS pitch is = 8 (mod 16) pixels for the narrow blocks (24 or 40), and the second row of a 2 x 16 block takes its
// columns rotated by 8: the 16 LDS rows of every ds_read_b128 lane group ({0-3, 12-15, 20-27}, ...) then stay distinct modulo
// 16, as 16 consecutive rows are.
//
// Global loads are buffer loads (a halo pixel outside the image, a weight row beyond N and the fetch past the last slab pass an
// offset beyond the resource: zeros, no branch), requested THREE stages (= one slab) before their LDS store -- a ring of three
// weight register sets, one per tap row, and the halo registers of the next slab -- and ISSUED between the matrix instructions
// of a stage, a few per step: the texture path moves 64 B / clock, and the 13 loads of four waves in one burst before the
// barrier held every wave for ~800 cycles (scripts/exp/conv_timeline.hip: clock marks per wave and stage, and the residency of
// the workgroups over the launch).  Within a stage the LDS fragments of step i + 1 are requested before the products of step i.
// Algorithmic bytes = 2 B H W (C + N) + 18 N C;  flops = 18 B H W C N.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include <mdetr_wave.h>

#include "conv3x3.h"
#include "mdetr_tune.h"
#include "msda.h"       // profile scopes

namespace mdetr {
namespace {

constexpr int kWavesC = 4;               // waves per workgroup: 128 output pixels
constexpr int kSlab = 64;                // input channels per LDS slab
constexpr int kPad = kSlab + 8;          // 72 bf16 per LDS row
constexpr int kThreads = kWavesC * 64;

struct ConvDims {
    int B, H, W, C, N;
    int tiles_x, tiles_y;                // column / row tiles per image
    int tiles;                           // B * tiles_x * tiles_y
    int ngroups;                         // output-channel groups of NB*32
    int xcd_per;                         // 8 / ngroups when that is whole (XCD-aware numbering below), else 0
    int mirror;                          // taps read mirrored: w[n][2 - t][2 - s][k] (the input gradient: no mirrored weight copy)
    int relu;
};

// Tile of a workgroup: its four waves' blocks (32 / WC rows x WC columns each) GC side by side, 4 / GC one below the other.
template <int WC, int GC> struct Tile {
    static constexpr int WR = 32 / WC, GR = kWavesC / GC;
    static constexpr int rows = GR * WR, cols = GC * WC;
    static constexpr int halo_h = rows + 2, halo_w = cols + 2;
    // LDS pitch of a halo row in pixels: any for 32-wide blocks; = 8 (mod 16) for the narrow ones (see the file comment)
    static constexpr int pitch = WC == 32 ? halo_w : (halo_w <= 24 ? 24 : 40);
    static constexpr int halo_pixels = halo_h * halo_w;
    static constexpr size_t lds(int nb) { return static_cast<size_t>(halo_h) * pitch * kPad * 2 + 3 * nb * 32 * kPad * 2 + nb * 32 * 4; }
};

// scripts/exp/conv_timeline.hip compiles this file with the macro set: clock marks of a few workgroups, per wave and stage
#ifdef MDETR_CONV3X3_TIMELINE
__device__ long long conv_tl[16][4][128];
__device__ long long conv_span[4096][4];                  // per workgroup: start, end, HW_ID, XCC_ID
#define TL_MARK(i) do { if (lane == 0 && tl_slot >= 0 && (i) < 128) conv_tl[tl_slot][wave][(i)] = clock64(); } while (0)
#else
#define TL_MARK(i) do { } while (0)
#endif

template <int NB, int WC, int GC>
__global__ __launch_bounds__(kThreads)
void conv3x3_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w, const float *__restrict__ shift,
                    __bf16 *__restrict__ y, const ConvDims d, const __bf16 *__restrict__ mask)
{
    using TL = Tile<WC, GC>;
    constexpr int WR = TL::WR, HWL = TL::pitch, kHaloH = TL::halo_h, kHaloW = TL::halo_w;
    MDETR_DYNAMIC_LDS(unsigned char, conv_smem);
    __bf16 *halo = reinterpret_cast<__bf16 *>(conv_smem);                    // [kHaloH][HWL][kPad]
    __bf16 *wts = halo + kHaloH * HWL * kPad;                               // [3 taps][NB*32][kPad]
    float *shift_s = reinterpret_cast<float *>(wts + 3 * NB * 32 * kPad);    // [NB*32]
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6), half = lane >> 5, col = lane & 31;
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
    const int r0 = ty * TL::rows, c0 = tx * TL::cols, n0 = group * NB * 32;
    const mdetr_rsrc xr = make_rsrc(x + static_cast<int64_t>(b) * d.H * d.W * d.C, static_cast<unsigned>(d.H * d.W) * static_cast<unsigned>(d.C * 2));
    const mdetr_rsrc wr = make_rsrc(w, static_cast<unsigned>(d.N * 9) * static_cast<unsigned>(d.C * 2));

    // this lane's pixel within the tile, and whether the wave's block meets the image at all (uniform)
    const int gr = wave / GC, gc = wave - gr * GC;
    const int pin = WC == 16 ? ((col & 15) + 8 * (col >> 4)) & 15 : col % WC;
    const int prow = gr * WR + col / WC, pcol = gc * WC + pin;
    const bool active = r0 + gr * WR < d.H && c0 + gc * WC < d.W;

    for (int i = threadIdx.x; i < NB * 32; i += kThreads) shift_s[i] = (shift && n0 + i < d.N) ? shift[n0 + i] : 0.f;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;

    // Staging: thread p + j * 256 moves 16-byte piece p & 7 of LDS row p >> 3 (8 consecutive lanes = the 128 contiguous bytes of one
    // row).  Per-thread byte offsets of slab 0 / tap row 0 are fixed; slab and tap row enter through the scalar offset.
    constexpr int WP = 3 * NB * 32 * (kSlab / 8) / kThreads;                // weight pieces per thread and stage (3 NB)
    constexpr int HP = (kHaloH * kHaloW * (kSlab / 8) + kThreads - 1) / kThreads;     // halo pieces per thread and slab (7 of a 4 x 32 tile)
    unsigned w_off[WP], h_off[HP];
    int h_lds[HP];
#pragma unroll
    for (int j = 0; j < WP; ++j) {
        const int p = threadIdx.x + j * kThreads, piece = p & 7, row = p >> 3;             // row = s * NB*32 + n
        const int s = row / (NB * 32), n = row - s * (NB * 32);
        w_off[j] = n0 + n < d.N ? static_cast<unsigned>(((n0 + n) * 9 + (d.mirror ? 2 - s : s)) * d.C + piece * 8) * 2u : kRsrcOob;
    }
#pragma unroll
    for (int j = 0; j < HP; ++j) {
        const int p = threadIdx.x + j * kThreads, piece = p & 7, pix = p >> 3;
        const int hr = pix / kHaloW, hc = pix - hr * kHaloW;
        const int r = r0 + hr - 1, c = c0 + hc - 1;
        const bool in = pix < kHaloH * kHaloW && r >= 0 && r < d.H && c >= 0 && c < d.W;
        h_off[j] = in ? static_cast<unsigned>((r * d.W + c) * d.C + piece * 8) * 2u : kRsrcOob;
        h_lds[j] = pix < kHaloH * kHaloW ? (hr * HWL + hc) * kPad + piece * 8 : -1;
    }
    bf16x8 wreg[3][WP], hreg[HP];
    // one load of the ring: piece j of tap row `slot` (j < WP) or of the halo (j >= WP) of the slab at channel k0
    auto fetch_piece = [&](auto slot, int j, int k0, bool valid, bool with_halo) {
        if (j < WP) {
            const unsigned so = static_cast<unsigned>(((d.mirror ? 2 - slot.value : slot.value) * 3 * d.C + k0) * 2);
            wreg[slot.value][j] = rsrc_load_bf16x8(wr, valid ? w_off[j] : kRsrcOob, so);
        } else if (with_halo && j - WP < HP) {
            hreg[j - WP] = rsrc_load_bf16x8(xr, valid ? h_off[j - WP] : kRsrcOob, static_cast<unsigned>(k0 * 2));
        }
    };
    auto fetch_all = [&](auto slot, int k0, bool with_halo) {
#pragma unroll
        for (int j = 0; j < WP + HP; ++j) fetch_piece(slot, j, k0, true, with_halo);
    };
    auto store_w = [&](auto slot) {
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int p = threadIdx.x + j * kThreads;
            *reinterpret_cast<bf16x8 *>(wts + (p >> 3) * kPad + (p & 7) * 8) = wreg[slot.value][j];
        }
    };
    auto store_h = [&]() {
#pragma unroll
        for (int j = 0; j < HP; ++j)                                        // (only the last piece index is ragged: 1632 pieces on 256 threads)
            if ((j + 1) * kThreads <= kHaloH * kHaloW * (kSlab / 8) || h_lds[j] >= 0) *reinterpret_cast<bf16x8 *>(halo + h_lds[j]) = hreg[j];
    };
    // The products of tap row `slot` (12 steps: 3 taps x 4 k-steps of 16), with the ring's loads for the NEXT slab issued between
    // them, a few per step: the texture path moves 64 B / clock -- the 13 loads of four waves issued in one burst before the barrier
    // held every wave for ~800 cycles (scripts/exp/conv_timeline.hip) -- and the matrix pipes run meanwhile.  Fragments of step
    // i + 1 are requested from LDS before the matrix instructions of step i.
    constexpr int kSteps = 3 * (kSlab / 16);
    auto products = [&](auto slot, int kn, bool more, bool with_halo) {
        constexpr int tr = slot.value;
        constexpr int per_step = (WP + HP + kSteps - 1) / kSteps;           // loads per step when the halo rides along
        const __bf16 *hp0 = halo + ((prow + tr) * HWL + pcol) * kPad + half * 8;
        const __bf16 *wp0 = wts + col * kPad + half * 8;
        bf16x8 xf[2], wf[2][NB];
        auto frags = [&](int i, int buf) {
            const int s = i / (kSlab / 16), ks = i % (kSlab / 16);
            xf[buf] = *reinterpret_cast<const bf16x8 *>(hp0 + s * kPad + ks * 16);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wf[buf][nb] = *reinterpret_cast<const bf16x8 *>(wp0 + (s * NB * 32 + nb * 32) * kPad + ks * 16);
        };
        if (active) {                                                       // (ONE uniform branch: straight-line code inside, or the wait counters turn conservative)
            frags(0, 0);
#pragma unroll
            for (int i = 0; i < kSteps; ++i) {
                if (i + 1 < kSteps) frags(i + 1, (i + 1) & 1);
#pragma unroll
                for (int j = i * per_step; j < (i + 1) * per_step; ++j) fetch_piece(slot, j, kn, more, with_halo);
                __builtin_amdgcn_sched_barrier(0);                          // (the scheduler otherwise sinks the requests to their first use: read, wait, multiply)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_bf16(wf[i & 1][nb], xf[i & 1], acc[nb]);      // Y^T[n][pixel]
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < kSteps * per_step; ++j) fetch_piece(slot, j, kn, more, with_halo);
        }
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;

    const int slabs = d.C / kSlab;
#ifdef MDETR_CONV3X3_TIMELINE
    const int tl_slot = blockIdx.x % 67 == 0 && blockIdx.x / 67 < 16 ? blockIdx.x / 67 : -1;
#endif
    TL_MARK(0);
#ifdef MDETR_CONV3X3_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        conv_span[blockIdx.x][0] = wall_clock64();
        conv_span[blockIdx.x][2] = __builtin_amdgcn_s_getreg(63492);
        conv_span[blockIdx.x][3] = __builtin_amdgcn_s_getreg(63508);
    }
#endif
    fetch_all(T0{}, 0, true);
    fetch_all(T1{}, 0, false);
    fetch_all(T2{}, 0, false);
    for (int sl = 0; sl < slabs; ++sl) {
        const bool more = sl + 1 < slabs;                                   // (past the last slab the ring fetches zeros: no branch around loads)
        const int kn = (sl + 1) * kSlab;
        TL_MARK(1 + sl * 15);
        __syncthreads();                                                    // the previous stage's LDS reads are done
        TL_MARK(2 + sl * 15);
        store_h();
        store_w(T0{});
        TL_MARK(3 + sl * 15);
        __syncthreads();
        TL_MARK(4 + sl * 15);
        products(T0{}, kn, more, true);
        TL_MARK(5 + sl * 15);
        __syncthreads();
        TL_MARK(6 + sl * 15);
        store_w(T1{});
        TL_MARK(7 + sl * 15);
        __syncthreads();
        TL_MARK(8 + sl * 15);
        products(T1{}, kn, more, false);
        TL_MARK(9 + sl * 15);
        __syncthreads();
        TL_MARK(10 + sl * 15);
        store_w(T2{});
        TL_MARK(11 + sl * 15);
        __syncthreads();
        TL_MARK(12 + sl * 15);
        products(T2{}, kn, more, false);
        TL_MARK(13 + sl * 15);
    }
    TL_MARK(127);
#ifdef MDETR_CONV3X3_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < 4096) conv_span[blockIdx.x][1] = wall_clock64();
#endif

    // ---- epilogue: lane = pixel; register quad g of block nb = channels 32 nb + 8 g + 4 half + 0..3
    const int r = r0 + prow, c = c0 + pcol;
    if (r < d.H && c < d.W) {
        const int64_t at = ((static_cast<int64_t>(b) * d.H + r) * d.W + c) * d.N + n0 + 4 * half;
        __bf16 *yp = y + at;
        // mask (the input gradient of a convolution whose INPUT is a ReLU output): the result is zeroed where mask <= 0 -- that
        // ReLU's backward, applied where its gradient leaves the chip (linear.ReluToken)
        bf16x4 mk[NB][4];
        if (mask) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (n0 + nb * 32 + 8 * g + 4 * half < d.N) mk[nb][g] = *reinterpret_cast<const bf16x4 *>(mask + at + nb * 32 + 8 * g);
        }
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
                        if (d.relu) v = v > 0.f ? v : 0.f;
                        if (mask && !(static_cast<float>(mk[nb][g][i]) > 0.f)) v = 0.f;
                        o[i] = static_cast<__bf16>(v);
                    }
                    *reinterpret_cast<bf16x4 *>(yp + nb * 32 + 8 * g) = o;
                }
            }
    }
}

template <int NB, int WC, int GC>
hipError_t launch(const void *x, const void *w, const float *shift, void *y, const ConvDims &d, hipStream_t st, const void *mask)
{
    using TL = Tile<WC, GC>;
    constexpr size_t lds = TL::lds(NB);
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    auto kern = conv3x3_kernel<NB, WC, GC>;
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
    g.tiles_x = (d.W + TL::cols - 1) / TL::cols;
    g.tiles_y = (d.H + TL::rows - 1) / TL::rows;
    g.tiles = d.B * g.tiles_x * g.tiles_y;
    g.ngroups = (d.N + NB * 32 - 1) / (NB * 32);
    g.xcd_per = (g.ngroups <= 8 && 8 % g.ngroups == 0) ? 8 / g.ngroups : 0;
    const int64_t blocks = g.xcd_per ? 8ll * ((g.tiles + g.xcd_per - 1) / g.xcd_per) : static_cast<int64_t>(g.tiles) * g.ngroups;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), lds, st, static_cast<const __bf16 *>(x),
                       static_cast<const __bf16 *>(w), shift, static_cast<__bf16 *>(y), g, static_cast<const __bf16 *>(mask));
    return hipGetLastError();
}

template <int WC, int GC>
hipError_t by_width(int nb, const void *x, const void *w, const float *shift, void *y, const ConvDims &d, hipStream_t st, const void *mask)
{
    if (nb == 4) return launch<4, WC, GC>(x, w, shift, y, d, st, mask);
    if (nb == 2) return launch<2, WC, GC>(x, w, shift, y, d, st, mask);
    return launch<1, WC, GC>(x, w, shift, y, d, st, mask);
}

// The tile shapes built, and what a choice costs.  A workgroup's duration hardly depends on what shares its CU (one alone at 64
// channels: 14 us at layer3, two per CU: 16 us, scripts/exp/conv_timeline.hip) -- a kernel takes ROUNDS of workgroup durations, and
// the dispatcher refills a CU as its workgroups end: 576 workgroups on 512 places ran 13 us at 512 resident, then 17 us at 64.
// So: as few rounds as possible first (workgroups <= places: one), the per-round cost of the channel width second.
struct Shape { int wc, gc, rows, cols; };
constexpr Shape kShapes[] = {{32, 1, 4, 32}, {16, 1, 8, 16}, {16, 2, 4, 32}, {8, 4, 4, 32}, {8, 2, 8, 16}};

struct Choice { int shape, nb; };

inline Choice choose(int B, int H, int W, int N)
{
    const int nb_cap = N >= 128 ? 4 : (N >= 64 ? 2 : 1);
    const int fnb = tune_int("conv3x3_nb", 0), ftile = tune_int("conv3x3_tile", 0);        // tests / A-B runs: nb in {1, 2, 4}; tile = 10 wc + gc
    Choice best{0, 1};
    double best_cost = 1e30;
    for (int si = 0; si < static_cast<int>(sizeof(kShapes) / sizeof(kShapes[0])); ++si) {
        const Shape &sh = kShapes[si];
        if (ftile > 0 && ftile != sh.wc * 10 + sh.gc) continue;
        const int wr = 32 / sh.wc;
        const int64_t tiles = static_cast<int64_t>(B) * ((H + sh.rows - 1) / sh.rows) * ((W + sh.cols - 1) / sh.cols);
        // wave blocks that meet the image (the others skip the products), per image
        const int blocks = ((W + sh.wc - 1) / sh.wc) * ((H + wr - 1) / wr);
        for (int nb = 1; nb <= nb_cap; nb *= 2) {
            if ((fnb == 1 || fnb == 2 || fnb == 4) && nb != (fnb > nb_cap ? nb_cap : fnb)) continue;
            const int64_t wgs = tiles * ((N + nb * 32 - 1) / (nb * 32));
            const int places = 256 * (nb == 1 ? 3 : (nb == 2 ? 2 : 1));       // workgroups resident on the device (LDS: 43 / 57-69 / 85-97 KB)
            const double rounds = wgs <= places ? 1.0 : (wgs <= 3 * places ? static_cast<double>((wgs + places - 1) / places) : static_cast<double>(wgs) / places + 0.5);
            const double per_round = nb == 1 ? 1.0 : (nb == 2 ? 1.28 : 1.7);  // measured workgroup durations per stage, relative
            const double cost = rounds * per_round * (1.0 + 1e-3 * blocks) * (1.0 + 1e-5 * (32 - sh.wc) + 1e-6 * sh.gc);      // ties: fewer live blocks, then the wider block
            if (cost < best_cost) { best_cost = cost; best = Choice{si, nb}; }
        }
    }
    return best;
}

}  // namespace

bool conv3x3_supported(int B, int H, int W, int C, int N, const void *x, const void *w, const void *y)
{
    const auto al = [](const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    return B > 0 && H > 0 && W > 0 && C > 0 && C % 64 == 0 && N > 0 && N % 32 == 0 && al(x, 16) && al(w, 16) && al(y, 8) &&
           static_cast<int64_t>(B) * ((H + 3) / 4) * ((W + 15) / 16) * ((N + 31) / 32) < (1ll << 30) &&
           static_cast<int64_t>(H) * W * C < (1ll << 30) && static_cast<int64_t>(N) * 9 * C < (1ll << 30);      // (32-bit byte offsets of one image / the weight)
}

int conv3x3_plan(int B, int H, int W, int N)
{
    const Choice c = choose(B, H, W, N);
    return kShapes[c.shape].wc * 100 + kShapes[c.shape].gc * 10 + c.nb;
}

hipError_t conv3x3_launch(const void *x, const void *w, const float *shift, void *y, int B, int H, int W, int C, int N, bool relu,
                          hipStream_t st, bool mirror, const void *mask)
{
    ConvDims d{B, H, W, C, N, 0, 0, 0, 0, 0, mirror ? 1 : 0, relu ? 1 : 0};
    ProfileScope prof(9, conv_mflop(static_cast<int64_t>(B) * H * W, static_cast<int64_t>(C) * N * 9), st, 18.0 * B * H * W * C * N / 1e6,
                      (2.0 * B * H * W * (C + N) + 18.0 * N * C) / 1e3);
    const Choice c = choose(B, H, W, N);
    switch (c.shape) {
    case 1: return by_width<16, 1>(c.nb, x, w, shift, y, d, st, mask);
    case 2: return by_width<16, 2>(c.nb, x, w, shift, y, d, st, mask);
    case 3: return by_width<8, 4>(c.nb, x, w, shift, y, d, st, mask);
    case 4: return by_width<8, 2>(c.nb, x, w, shift, y, d, st, mask);
    default: return by_width<32, 1>(c.nb, x, w, shift, y, d, st, mask);
    }
}

}  // namespace mdetr
