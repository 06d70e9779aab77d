// monodetr_amd/csrc/tgemm.hip -- y[T, N] = epilogue(a[T, K] op(w)) in bf16 on the matrix cores, for EVERY token-wise product of the
// training iteration: the backbone's 1x1 convolutions (torchvision Bottleneck.conv1 / conv3 / downsample behind
// lib/models/monodetr/backbone.py:93-106), the pyramid's input projections (monodetr.py:77-99), the deformable-attention
// projections (ops/modules/ms_deform_attn.py:94-102), the encoder's / decoder's FFNs and heads
// (depthaware_transformer.py:328-353, :431-435) -- forward (NT: w = W[N][K]) and input gradient (NN: w = W[K][N], the
// parameter as it lies in memory, transposed 8 x 8 in registers on its way into LDS).  The reference leaves all of these
// to cuBLAS / cuDNN followed by separate elementwise passes; here the tail of each product runs where its output tile
// leaves the chip:
//     y = dropout(relu(acc + bias + residual))        (each part optional; residual may BE y: beta = 1 accumulation)
// i.e. folded-BN shift + ReLU (conv1), shift + identity + ReLU (conv3), bias + ReLU + Dropout (FFN linear1), the
// residual-path gradient summed inside the input-gradient product (no 42 MB elementwise add).
//
// Tiling: a workgroup = 4 waves (2 x 2) owns BM tokens x BN features (128 x 128 down to 64 x 64, chosen by the
// launcher so that >= 2 workgroups per CU exist); the contraction runs in slabs of 64 through two LDS buffers per
// operand (rows padded to 72 bf16 = 36 dwords: the 16 rows of a ds_read_b128 lane group fall on 16 distinct 4-bank
// slots); global loads are register-staged two slabs ahead (the slab after next is in flight while this one's products
// issue, written to LDS after the barrier); products are issued transposed, D^T[n][token] += W[n][k] X[token][k]
// (v_mfma_f32_32x32x16_bf16, fragment conventions of mdetr_wave.h).  After the last slab the fp32 tile is parked in the
// (now free) LDS and leaves in whole rows: a thread owns 8 consecutive features of a token -- bias, residual (a 16-byte
// coalesced load issued before the tile is parked), ReLU, dropout, ONE rounding, a 16-byte store.
// Workgroup ids go round-robin over the 8 XCDs; the column tiles of one token range are ids 8 apart (same XCD, adjacent
// dispatch slots), so the second reader of an input tile finds it in that XCD's L2.
// Algorithmic bytes = 2 (T K + T N [+ T N residual]) + 2 N K; flops = 2 T N K.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mdetr_wave.h>

#include "add_ln_math.h"
#include "mdetr_transpose.h"
#include "msda.h"       // profile scopes
#include "tgemm.h"

namespace mdetr {
namespace {

constexpr int kThreadsT = 256;
constexpr int kBK = 64;                    // contraction values per slab
constexpr int kLd = kBK + 8;               // LDS row of a slab: 72 bf16 = 36 dwords
constexpr int kCPad = 4;                   // fp32 output tile: rows of BN + 4 floats (8 consecutive rows x 16 bytes: 8 distinct bank quads)

struct TgemmArgs {
    const __bf16 *a, *w, *res;
    const void *bias;
    void *y;
    int64_t T, lda, ldw, ldr, ldy;
    int N, K, gx, ny, flags;
    uint32_t thresh;
    float keep_scale;
    uint64_t seed;
    const uint64_t *seed_dev;
};

template <int BM, int BN>
constexpr size_t tgemm_lds()
{
    constexpr size_t slabs = static_cast<size_t>(2) * (BM + BN) * kLd * 2, tile = static_cast<size_t>(BM) * (BN + kCPad) * 4;
    return slabs > tile ? slabs : tile;
}

// PF = slabs of global loads in flight beyond the one being written to LDS (register sets)
template <int BM, int BN, bool NN, int PF>
__global__ __launch_bounds__(kThreadsT, 2)
void tgemm_kernel(const TgemmArgs g)
{
    constexpr int TM = BM / 64, TN = BN / 64;                    // 32 x 32 blocks of a wave along tokens / features
    constexpr int XCH = BM * 8 / kThreadsT;                      // 16-byte pieces of an input slab per thread
    constexpr int WCH = NN ? 8 : BN * 8 / kThreadsT;             // weight slab: pieces per thread (NT) / the 8 rows of one 8 x 8 block (NN)
    MDETR_DYNAMIC_LDS(unsigned char, tg_smem);
    __bf16 *Xs = reinterpret_cast<__bf16 *>(tg_smem);            // [2][BM][kLd]
    __bf16 *Ws = Xs + 2 * BM * kLd;                              // [2][BN][kLd]
    float *Cs = reinterpret_cast<float *>(tg_smem);              // [BM][BN + kCPad], after the last slab
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave & 1, wn = wave >> 1;
    const int id = blockIdx.x, grp = id >> 3;
    const int col = grp % g.ny, bx = (grp / g.ny) * 8 + (id & 7);
    const int64_t m0 = static_cast<int64_t>(bx) * BM;
    if (m0 >= g.T) return;                                       // (gx was rounded up to a multiple of 8)
    const int n0 = col * BN;
    const int KT = (g.K + kBK - 1) / kBK;

    bf16x8 zero8;
#pragma unroll
    for (int i = 0; i < 8; ++i) zero8[i] = static_cast<__bf16>(0.f);

    // ---- staging maps.  Input slab: piece c = tid + 256 j -> row c / 8, 16 bytes at k = 8 (c % 8): 8 lanes cover one 128-byte
    // row segment.  Rows beyond T re-read row T - 1 (their outputs are never stored); pieces beyond K are zero.
    const __bf16 *xsrc[XCH];
    int xdst[XCH];
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
        const int c = tid + kThreadsT * j, row = c >> 3, piece = c & 7;
        const int64_t t = m0 + row, tc = t < g.T ? t : g.T - 1;
        xsrc[j] = g.a + tc * g.lda + piece * 8;
        xdst[j] = row * kLd + piece * 8;
    }
    const int xk = (tid & 7) * 8;                                // this thread's k offset inside a slab (the same for all its pieces)
    // Weight slab, NT (w = W[N][K]): the same map over BN rows; rows beyond N re-read row N - 1 (columns never stored).
    // NN (w = W[K][N]): thread b < BN owns the 8 x 8 block (k-block b & 7, n-block b >> 3): 8 loads of 16 bytes (8 k-rows, 8
    // consecutive n), transposed in registers into 8 LDS rows (n) of 8 k -- the 8 lanes of a ds_write_b128 group hold the 8
    // k-blocks of one n-row: 128 contiguous bytes, conflict-free.
    const __bf16 *wsrc[NN ? 1 : WCH];
    int wdst[NN ? 1 : WCH];
    bool wlive = true;
    int wkb = 0;
    if (NN) {
        const int kb = tid & 7, nb = tid >> 3;
        wlive = tid < BN && n0 + nb * 8 < g.N;                   // (N % 8 == 0: a block's 8 columns are in or out together)
        wkb = kb * 8;
        wsrc[0] = g.w + static_cast<int64_t>(kb * 8) * g.ldw + (wlive ? n0 + nb * 8 : 0);
        wdst[0] = nb * 8 * kLd + kb * 8;
    } else {
#pragma unroll
        for (int j = 0; j < (NN ? 1 : WCH); ++j) {
            const int c = tid + kThreadsT * j, row = c >> 3, piece = c & 7;
            const int n = n0 + row < g.N ? n0 + row : g.N - 1;
            wsrc[j] = g.w + static_cast<int64_t>(n) * g.ldw + piece * 8;
            wdst[j] = row * kLd + piece * 8;
        }
    }

    bf16x8 xr[PF][XCH], wr[PF][WCH];
    auto fetch = [&](int kt, bf16x8 (&xs_)[XCH], bf16x8 (&ws_)[WCH]) __attribute__((always_inline)) {
        const int k0 = kt * kBK;
        const bool xin = k0 + xk < g.K;
#pragma unroll
        for (int j = 0; j < XCH; ++j) xs_[j] = xin ? *reinterpret_cast<const bf16x8 *>(xsrc[j] + k0) : zero8;
        if (NN) {
            if (tid < BN) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool in = wlive && k0 + wkb + i < g.K;
                    ws_[i] = in ? *reinterpret_cast<const bf16x8 *>(wsrc[0] + static_cast<int64_t>(k0 + i) * g.ldw) : zero8;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < WCH; ++j) ws_[j] = xin ? *reinterpret_cast<const bf16x8 *>(wsrc[NN ? 0 : j] + k0) : zero8;
        }
    };
    auto deposit = [&](int buf, const bf16x8 (&xs_)[XCH], const bf16x8 (&ws_)[WCH]) __attribute__((always_inline)) {
        __bf16 *xb = Xs + buf * BM * kLd, *wb = Ws + buf * BN * kLd;
#pragma unroll
        for (int j = 0; j < XCH; ++j) *reinterpret_cast<bf16x8 *>(xb + xdst[j]) = xs_[j];
        if (NN) {
            if (tid < BN) {
                bf16x8 in[8], tr[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) in[i] = ws_[i];
                transpose8x8(in, tr);
#pragma unroll
                for (int q = 0; q < 8; ++q) *reinterpret_cast<bf16x8 *>(wb + wdst[0] + q * kLd) = tr[q];
            }
        } else {
#pragma unroll
            for (int j = 0; j < WCH; ++j) *reinterpret_cast<bf16x8 *>(wb + wdst[NN ? 0 : j]) = ws_[j];
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a_ = 0; a_ < TN; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < TM; ++b_)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a_][b_][i] = 0.f;

    auto products = [&](int buf) __attribute__((always_inline)) {
        const __bf16 *xl = Xs + (buf * BM + wm * (BM / 2) + l31) * kLd + half * 8;     // + 32 tm rows, + 16 ks
        const __bf16 *wl = Ws + (buf * BN + wn * (BN / 2) + l31) * kLd + half * 8;
#pragma unroll
        for (int ks = 0; ks < kBK / 16; ++ks) {
            bf16x8 xf[TM], wf[TN];
#pragma unroll
            for (int b_ = 0; b_ < TM; ++b_) xf[b_] = *reinterpret_cast<const bf16x8 *>(xl + b_ * 32 * kLd + ks * 16);
#pragma unroll
            for (int a_ = 0; a_ < TN; ++a_) wf[a_] = *reinterpret_cast<const bf16x8 *>(wl + a_ * 32 * kLd + ks * 16);
#pragma unroll
            for (int a_ = 0; a_ < TN; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < TM; ++b_) acc[a_][b_] = mfma_bf16(wf[a_], xf[b_], acc[a_][b_]);      // D^T[n][token]
        }
    };

    // ---- slab pipeline: slab s + 1 is written to LDS after the barrier that freed its buffer, slabs s + 2 .. s + 1 + PF are in
    // flight in registers while slab s's products issue
    fetch(0, xr[0], wr[0]);
    deposit(0, xr[0], wr[0]);
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (1 + p < KT) fetch(1 + p, xr[p], wr[p]);
    __syncthreads();
    for (int kt = 0; kt < KT; kt += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int s = kt + p;
            if (s < KT) {                                        // (uniform)
                if (s + 1 < KT) deposit((s + 1) & 1, xr[p], wr[p]);
                if (s + 1 + PF < KT) fetch(s + 1 + PF, xr[p], wr[p]);
                products(s & 1);
                __syncthreads();
            }
        }
    }

    // ---- epilogue.  piece c = tid + 256 j of the output tile: row c / (BN / 8), 8 features at 8 (c % (BN / 8)); 256 % (BN / 8) == 0,
    // so a thread's feature group -- its bias -- is the same for all its pieces.
    constexpr int PR = BN / 8, NP = BM * PR / kThreadsT;
    const int pc = tid % PR, n = n0 + pc * 8;
    const bool ncol = n < g.N;                                   // N % 8 == 0
    bf16x8 rr[NP];
    if (g.res) {                                                 // the residual tile: requested before the accumulators are parked
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int row = (tid + kThreadsT * j) / PR;
            const int64_t t = m0 + row;
            rr[j] = (ncol && t < g.T) ? *reinterpret_cast<const bf16x8 *>(g.res + t * g.ldr + n) : zero8;
        }
    }
    float bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bv[i] = 0.f;
    if (g.bias && ncol) {
        if (g.flags & kTgemmBiasF32) {
#pragma unroll
            for (int i = 0; i < 8; ++i) bv[i] = static_cast<const float *>(g.bias)[n + i];
        } else {
            const bf16x8 b8 = *reinterpret_cast<const bf16x8 *>(static_cast<const __bf16 *>(g.bias) + n);
#pragma unroll
            for (int i = 0; i < 8; ++i) bv[i] = static_cast<float>(b8[i]);
        }
    }
    // accumulator register 4 q + i of lane l: feature 8 q + 4 (l >> 5) + i, token l & 31 of its 32 x 32 block
#pragma unroll
    for (int a_ = 0; a_ < TN; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < TM; ++b_) {
            float *cr = Cs + (wm * (BM / 2) + b_ * 32 + l31) * (BN + kCPad) + wn * (BN / 2) + a_ * 32 + 4 * half;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
                v.x = acc[a_][b_][4 * q]; v.y = acc[a_][b_][4 * q + 1]; v.z = acc[a_][b_][4 * q + 2]; v.w = acc[a_][b_][4 * q + 3];
                *reinterpret_cast<f32x4 *>(cr + 8 * q) = v;
            }
        }
    __syncthreads();
    const bool relu = (g.flags & kTgemmRelu) != 0;
    const uint64_t sd = g.thresh ? g.seed + (g.seed_dev ? *g.seed_dev : 0ull) : 0ull;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int row = (tid + kThreadsT * j) / PR;
        const int64_t t = m0 + row;
        if (!ncol || t >= g.T) continue;
        const float *cr = Cs + row * (BN + kCPad) + pc * 8;
        const f32x4 c0 = *reinterpret_cast<const f32x4 *>(cr), c1 = *reinterpret_cast<const f32x4 *>(cr + 4);
        float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float f = v[i] + bv[i];
            if (g.res) f += static_cast<float>(rr[j][i]);
            if (relu) f = f < 0.f ? 0.f : f;                     // (NaN passes through, as clamp_min does)
            if (g.thresh) f = ln_hash(sd, static_cast<uint64_t>(t) * static_cast<uint64_t>(g.N) + static_cast<uint64_t>(n + i)) >= g.thresh ? f * g.keep_scale : 0.f;
            v[i] = f;
        }
        if (g.flags & kTgemmOutF32) {
            float *yp = static_cast<float *>(g.y) + t * g.ldy + n;
            f32x4 o0, o1;
            o0.x = v[0]; o0.y = v[1]; o0.z = v[2]; o0.w = v[3]; o1.x = v[4]; o1.y = v[5]; o1.z = v[6]; o1.w = v[7];
            *reinterpret_cast<f32x4 *>(yp) = o0;
            *reinterpret_cast<f32x4 *>(yp + 4) = o1;
        } else {
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = static_cast<__bf16>(v[i]);
            *reinterpret_cast<bf16x8 *>(static_cast<__bf16 *>(g.y) + t * g.ldy + n) = o;
        }
    }
}

template <int BM, int BN, bool NN, int PF>
hipError_t launch_tile(TgemmArgs g, hipStream_t st)
{
    constexpr size_t lds = tgemm_lds<BM, BN>();
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    auto kern = tgemm_kernel<BM, BN, NN, PF>;
    static bool attr_set[64] = {};                               // the attribute is per device: one process may drive several GPUs
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = -1;
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    const int64_t tiles_m = (g.T + BM - 1) / BM;
    g.ny = (g.N + BN - 1) / BN;
    g.gx = static_cast<int>((tiles_m + 7) / 8 * 8);              // whole rounds over the XCDs (idle workgroups leave at once)
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(static_cast<int64_t>(g.gx) * g.ny)), dim3(kThreadsT), lds, st, g);
    return hipGetLastError();
}

template <bool NN, int PF>
hipError_t launch_any(const TgemmArgs &g, int bm, int bn, hipStream_t st)
{
    if (bm == 128 && bn == 128) return launch_tile<128, 128, NN, PF>(g, st);
    if (bm == 128) return launch_tile<128, 64, NN, PF>(g, st);
    if (bn == 128) return launch_tile<64, 128, NN, PF>(g, st);
    return launch_tile<64, 64, NN, PF>(g, st);
}

}  // namespace

bool tgemm_supported(const TgemmProblem &p)
{
    const auto al = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool nn = (p.flags & kTgemmNN) != 0;
    return p.T > 0 && p.T < (1ll << 31) - 256 && p.N > 0 && p.N % 8 == 0 && p.K > 0 && p.K % 8 == 0 && p.a && p.w && p.y && al(p.a) && al(p.w) &&
           al(p.y) && p.lda % 8 == 0 && p.lda >= p.K && p.ldw % 8 == 0 && p.ldw >= (nn ? p.N : p.K) &&
           p.ldy % ((p.flags & kTgemmOutF32) ? 4 : 8) == 0 && p.ldy >= p.N &&
           (!p.res || (al(p.res) && p.ldr % 8 == 0 && p.ldr >= p.N)) && (!p.bias || al(p.bias)) && p.dropout_p >= 0.f && p.dropout_p < 1.f;
}

hipError_t tgemm_launch(const TgemmProblem &p, hipStream_t st)
{
    TgemmArgs g;
    g.a = static_cast<const __bf16 *>(p.a); g.w = static_cast<const __bf16 *>(p.w); g.res = static_cast<const __bf16 *>(p.res);
    g.bias = p.bias; g.y = p.y;
    g.T = p.T; g.lda = p.lda; g.ldw = p.ldw; g.ldr = p.ldr; g.ldy = p.ldy;
    g.N = p.N; g.K = p.K; g.gx = g.ny = 0; g.flags = p.flags;
    g.thresh = p.dropout_p > 0.f ? ln_threshold(p.dropout_p) : 0u;
    g.keep_scale = p.dropout_p > 0.f ? 1.f / (1.f - p.dropout_p) : 1.f;
    g.seed = p.seed; g.seed_dev = p.seed_dev;
    // the largest tile that still gives every CU two workgroups (512); narrow outputs take 64-wide feature tiles
    int bm = 128, bn = p.N <= 64 ? 64 : 128;
    const auto wgs = [&](int m, int n) { return ((p.T + m - 1) / m) * ((p.N + n - 1) / n); };
    if (wgs(bm, bn) < 512) bm = 64;
    if (wgs(bm, bn) < 512 && bn == 128) bn = 64;
    if (const char *ev = getenv("MDETR_TGEMM_TILE")) {           // A/B runs: "128x64"
        int m = 0, n = 0;
        if (sscanf(ev, "%dx%d", &m, &n) == 2 && (m == 64 || m == 128) && (n == 64 || n == 128)) { bm = m; bn = n; }
    }
    int pf = 2;
    if (const char *ev = getenv("MDETR_TGEMM_PF")) pf = atoi(ev) == 1 ? 1 : 2;       // A/B runs: register sets in flight
    ProfileScope prof(10, conv_mflop(p.T, static_cast<int64_t>(p.N) * p.K), st);
    if (p.flags & kTgemmNN) return pf == 1 ? launch_any<true, 1>(g, bm, bn, st) : launch_any<true, 2>(g, bm, bn, st);
    return pf == 1 ? launch_any<false, 1>(g, bm, bn, st) : launch_any<false, 2>(g, bm, bn, st);
}

}  // namespace mdetr
