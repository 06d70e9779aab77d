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
#include "mdetr_tune.h"

namespace mdetr {
namespace {

constexpr int kThreadsT = 256;
constexpr int kBK = 64;                    // contraction values per slab
constexpr int kLd = kBK + 8;               // LDS row of a slab: 72 bf16 = 36 dwords
constexpr int kCPad = 4;                   // fp32 output tile: rows of BN + 4 floats (8 consecutive rows x 16 bytes: 8 distinct bank quads)

struct TgemmArgs {
    const __bf16 *a, *w, *res, *mask;
    const void *bias;
    void *y;
    int64_t T, lda, ldw, ldr, ldy, ldm;
    int N, K, gx, ny, flags, tiles_m;                       // tiles_m: row tiles (the host divides: a 64-bit division in the kernel's first microsecond otherwise)
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

// 4 k-rows x 8 columns of bf16 (in[i] = 8 consecutive n of k-row i) -> out[q] = the 4 k values of column q (8 bytes)
__device__ __forceinline__ void transpose4x8(const bf16x8 (&in)[4], unsigned (&out)[8][2])
{
    unsigned d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_memcpy(d[i], &in[i], 16);
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const unsigned lo = d[2 * m][q >> 1], hi = d[2 * m + 1][q >> 1];
            out[q][m] = (q & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        }
}

// PF = slabs of global loads in flight beyond the one being written to LDS (register sets).
// PERSISTENT workgroups: workgroup w walks the output tiles w, w + G, w + 2 G, ... (G = gridDim.x, a multiple of 8: a tile keeps
// its XCD) as ONE sequence of contraction slabs -- the loads of the next tile's first slabs are in flight while this tile's last
// products issue and its output leaves, so only a workgroup's very first slab pays an exposed memory latency.
// TAIL: 0 = bias (bf16) + ReLU, bf16 output;  1 = the same + a residual tile;  2 = everything (fp32 bias / output, dropout, residual);
//       3 = tail 1 (residual optional) times the sign mask of a second [T, N] tensor: y = mask > 0 ? acc + res : 0 -- the ReLU backward of
//           the layer's INPUT applied where the input gradient leaves the chip (the mask tile is requested once the accumulators are parked)
// NTH: threads of the workgroup -- 256 (waves 2 x 2) or 512 (2 x 4: the big tile with twice the waves in flight per CU)
// scripts/exp/tgemm_timeline.hip compiles this file with the macro set: clock marks of workgroup 0's first wave
#ifdef MDETR_TGEMM_TIMELINE
__device__ long long tgemm_tl[64];
#define TG_MARK(i) do { if (blockIdx.x == 3 && threadIdx.x == 0 && (i) < 64) tgemm_tl[(i)] = clock64(); } while (0)
#else
#define TG_MARK(i) do { } while (0)
#endif

template <int BM, int BN, bool NN, int PF, int TAIL, int NTH = 256>
__global__ __launch_bounds__(NTH, NTH == 512 ? 4 : 2)
void tgemm_kernel(const TgemmArgs g)
{
    TG_MARK(0);
    constexpr int kThreadsT = NTH;                               // (shadows the file-level default inside this kernel)
    constexpr int WNW = NTH / 128;                               // waves along the features (2 along the tokens)
    constexpr int TM = BM / 64, TN = BN / (32 * WNW);            // 32 x 32 blocks of a wave along tokens / features
    constexpr int XCH = BM * 8 / kThreadsT;                      // 16-byte pieces of an input slab per thread
    constexpr int WCH = NN ? 4 : BN * 8 / kThreadsT;             // weight slab: pieces per thread (NT) / the 4 k-rows of one 4 x 8 block (NN)
    constexpr int WTHR = NN ? 2 * BN : kThreadsT;                // threads that stage the weight slab (NN: 16 k-groups x BN / 8 column blocks)
    static_assert(TM >= 1 && TN >= 1 && XCH >= 1 && WCH >= 1 && WTHR <= kThreadsT, "tile / workgroup shape");
    MDETR_DYNAMIC_LDS(unsigned char, tg_smem);
    __bf16 *Xs = reinterpret_cast<__bf16 *>(tg_smem);            // [2][BM][kLd]
    __bf16 *Ws = Xs + 2 * BM * kLd;                              // [2][BN][kLd]
    float *Cs = reinterpret_cast<float *>(tg_smem);              // [BM][BN + kCPad], after a tile's last slab
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wm = wave & 1, wn = wave >> 1;                    // wave's place: 2 along the tokens, WNW along the features
    const int KT = (g.K + kBK - 1) / kBK;
    const int G = gridDim.x, vtiles = g.gx * g.ny;               // virtual tile ids (gx: row tiles rounded up to a multiple of 8)
    const int tiles_m = g.tiles_m;
    // virtual id v -> (column tile, row tile): the ny column tiles of one token range are ids 8 apart (same XCD, adjacent slots)
    auto tile_of = [&](int v, int &col, int64_t &m0) __attribute__((always_inline)) {
        const int grp = v >> 3;
        col = grp % g.ny;
        m0 = (static_cast<int64_t>(grp / g.ny) * 8 + (v & 7)) * BM;
    };
    auto next_tile = [&](int v) __attribute__((always_inline)) {  // the next virtual id >= v with a live row tile (or >= vtiles)
        for (; v < vtiles; v += G) {
            if ((v >> 3) / g.ny * 8 + (v & 7) < tiles_m) break;
        }
        return v;
    };

    // Loads go through buffer resources: a lane whose offset lies beyond the tensor (rows past T or N, pieces past K) receives
    // zeros -- no branch, no select on the data.  Per-lane offset = (row, piece) of the tile, scalar offset = the slab.
    const unsigned a_bytes = static_cast<unsigned>(((g.T - 1) * g.lda + g.K) * 2);
    const unsigned w_bytes = static_cast<unsigned>(NN ? ((static_cast<int64_t>(g.K) - 1) * g.ldw + g.N) * 2 : ((static_cast<int64_t>(g.N) - 1) * g.ldw + g.K) * 2);
    const mdetr_rsrc ar = make_rsrc(g.a, a_bytes), wr_ = make_rsrc(g.w, w_bytes);
    const int xk = (tid & 7) * 8;                                // NT maps: this thread's k offset inside a slab (the same for all its pieces)
    const int wkg = tid & 15, wnb = tid >> 4;                    // NN map: k-group (4 rows) and column block (8 columns) of this thread
    unsigned xoff[XCH], woff[NN ? 1 : WCH];                      // per-lane byte offsets of the FETCH cursor's tile
    int xdst[XCH], wdst[NN ? 1 : WCH];
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
        const int c = tid + kThreadsT * j;
        xdst[j] = (c >> 3) * kLd + (c & 7) * 8;
    }
    if (NN) {
        wdst[0] = wnb * 8 * kLd + wkg * 4;
    } else {
#pragma unroll
        for (int j = 0; j < (NN ? 1 : WCH); ++j) {
            const int c = tid + kThreadsT * j;
            wdst[j] = (c >> 3) * kLd + (c & 7) * 8;
        }
    }
    auto aim = [&](int v) __attribute__((always_inline)) {       // the fetch cursor moves to tile v
        int col; int64_t m0;
        tile_of(v, col, m0);
        const int n0 = col * BN;
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            const int c = tid + kThreadsT * j;
            const int64_t t = m0 + (c >> 3);
            xoff[j] = t < g.T ? (static_cast<unsigned>(t) * static_cast<unsigned>(g.lda) + (c & 7) * 8) * 2u : kRsrcOob;      // (T lda < 2^30: tgemm_supported)
        }
        if (NN) {
            woff[0] = (tid < WTHR && n0 + wnb * 8 < g.N) ? static_cast<unsigned>((static_cast<int64_t>(wkg * 4) * g.ldw + n0 + wnb * 8) * 2) : kRsrcOob;
        } else {
#pragma unroll
            for (int j = 0; j < (NN ? 1 : WCH); ++j) {
                const int c = tid + kThreadsT * j, row = n0 + (c >> 3);
                woff[j] = row < g.N ? (static_cast<unsigned>(row) * static_cast<unsigned>(g.ldw) + (c & 7) * 8) * 2u : kRsrcOob;
            }
        }
    };

    bf16x8 xr[PF][XCH], wr[PF][WCH];
    auto fetch = [&](int kt, bf16x8 (&xs_)[XCH], bf16x8 (&ws_)[WCH]) __attribute__((always_inline)) {
        const int k0 = kt * kBK;
        const bool xin = k0 + xk < g.K;                          // (only a ragged last slab has dead pieces)
#pragma unroll
        for (int j = 0; j < XCH; ++j) xs_[j] = rsrc_load_bf16x8(ar, xin ? xoff[j] : kRsrcOob, static_cast<unsigned>(k0 * 2));
        if (NN) {
            if (tid < WTHR) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    ws_[i] = rsrc_load_bf16x8(wr_, k0 + wkg * 4 + i < g.K ? woff[0] : kRsrcOob,
                                              static_cast<unsigned>((static_cast<int64_t>(k0 + i) * g.ldw) * 2));
            }
        } else {
#pragma unroll
            for (int j = 0; j < WCH; ++j) ws_[j] = rsrc_load_bf16x8(wr_, xin ? woff[NN ? 0 : j] : kRsrcOob, static_cast<unsigned>(k0 * 2));
        }
    };
    auto deposit = [&](int buf, const bf16x8 (&xs_)[XCH], const bf16x8 (&ws_)[WCH]) __attribute__((always_inline)) {
        __bf16 *xb = Xs + buf * BM * kLd, *wb = Ws + buf * BN * kLd;
#pragma unroll
        for (int j = 0; j < XCH; ++j) *reinterpret_cast<bf16x8 *>(xb + xdst[j]) = xs_[j];
        if (NN) {
            if (tid < WTHR) {
                bf16x8 in[4];
                unsigned tr[8][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) in[i] = ws_[i];
                transpose4x8(in, tr);
#pragma unroll
                for (int q = 0; q < 8; ++q) __builtin_memcpy(wb + wdst[0] + q * kLd, tr[q], 8);
            }
        } else {
#pragma unroll
            for (int j = 0; j < WCH; ++j) *reinterpret_cast<bf16x8 *>(wb + wdst[NN ? 0 : j]) = ws_[j];
        }
    };

    f32x16 acc[TN][TM];
    auto clear = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a_ = 0; a_ < TN; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < TM; ++b_)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a_][b_][i] = 0.f;
    };
    auto products = [&](int buf) __attribute__((always_inline)) {
        const __bf16 *xl = Xs + (buf * BM + wm * (BM / 2) + l31) * kLd + half * 8;     // + 32 tm rows, + 16 ks
        const __bf16 *wl = Ws + (buf * BN + wn * (BN / WNW) + l31) * kLd + half * 8;
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

    // ---- a tile's tail.  piece c = tid + 256 j of the output tile: row c / (BN / 8), 8 features at 8 (c % (BN / 8)); 256 % (BN / 8)
    // == 0, so a thread's feature group -- its bias -- is the same for all its pieces.  Straight-line code: rows beyond T and
    // feature groups beyond N carry an out-of-range buffer offset (loads give zero, stores are dropped).
    constexpr int PR = BN / 8, NP = BM * PR / kThreadsT;
    constexpr bool RES = TAIL != 0;
    constexpr bool MASK = TAIL == 3;
    const int pc = tid % PR;
    const float lo = (g.flags & kTgemmRelu) ? 0.f : -__builtin_inff();
    const bool out32 = TAIL == 2 && (g.flags & kTgemmOutF32) != 0;
    const bool has_res = TAIL == 1 || ((TAIL == 2 || TAIL == 3) && g.res != nullptr);
    const uint64_t sd = (TAIL == 2 && g.thresh) ? g.seed + (g.seed_dev ? *g.seed_dev : 0ull) : 0ull;
    const mdetr_rsrc yr = make_rsrc(g.y, static_cast<unsigned>(((g.T - 1) * g.ldy + g.N) * (out32 ? 4 : 2)));
    const mdetr_rsrc rr_ = make_rsrc(has_res ? static_cast<const void *>(g.res) : static_cast<const void *>(g.a),
                                     has_res ? static_cast<unsigned>(((g.T - 1) * g.ldr + g.N) * 2) : 0u);
    const mdetr_rsrc mr_ = make_rsrc(MASK ? static_cast<const void *>(g.mask) : static_cast<const void *>(g.a),
                                     MASK ? static_cast<unsigned>(((g.T - 1) * g.ldm + g.N) * 2) : 0u);
    constexpr int NPB = NP < 4 ? NP : 4;                         // pieces per batch of the tail (registers: 8 floats each)
    bf16x8 rq[RES ? NP : 1];
    bf16x8 mq[MASK ? NP : 1];
    float bv[8];
    unsigned yoff[NP];                                           // element offsets of the pieces (kRsrcOob: not stored)
    int64_t tail_m0 = 0;
    int tail_n = 0;
    auto aim_tail = [&](int v) __attribute__((always_inline)) { // where the tile goes; its residual tile and bias are requested
        int col;
        tile_of(v, col, tail_m0);
        tail_n = col * BN + pc * 8;
        const bool ncol = tail_n < g.N;                          // N % 8 == 0
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int64_t t = tail_m0 + (tid + kThreadsT * j) / PR;
            yoff[j] = (ncol && t < g.T) ? static_cast<unsigned>(t * g.ldy + tail_n) : kRsrcOob;
            if (RES && !MASK) rq[j] = rsrc_load_bf16x8(rr_, (has_res && ncol && t < g.T) ? static_cast<unsigned>((t * g.ldr + tail_n) * 2) : kRsrcOob, 0u);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) bv[i] = 0.f;
        if (g.bias && ncol) {
            if (TAIL == 2 && (g.flags & kTgemmBiasF32)) {
#pragma unroll
                for (int i = 0; i < 8; ++i) bv[i] = static_cast<const float *>(g.bias)[tail_n + i];
            } else {
                const bf16x8 b8 = *reinterpret_cast<const bf16x8 *>(static_cast<const __bf16 *>(g.bias) + tail_n);
#pragma unroll
                for (int i = 0; i < 8; ++i) bv[i] = static_cast<float>(b8[i]);
            }
        }
    };
    auto park = [&]() __attribute__((always_inline)) {          // accumulators -> LDS
        // accumulator register 4 q + i of lane l: feature 8 q + 4 (l >> 5) + i, token l & 31 of its 32 x 32 block
#pragma unroll
        for (int a_ = 0; a_ < TN; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < TM; ++b_) {
                float *cr = Cs + (wm * (BM / 2) + b_ * 32 + l31) * (BN + kCPad) + wn * (BN / WNW) + a_ * 32 + 4 * half;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 o;
                    o.x = acc[a_][b_][4 * q]; o.y = acc[a_][b_][4 * q + 1]; o.z = acc[a_][b_][4 * q + 2]; o.w = acc[a_][b_][4 * q + 3];
                    *reinterpret_cast<f32x4 *>(cr + 8 * q) = o;
                }
            }
    };
    auto aim_mask = [&]() __attribute__((always_inline)) {      // (after park: the accumulators' registers are free) the mask tile and, in
        const bool ncol = tail_n < g.N;                          // this tail, the residual tile are requested
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int64_t t = tail_m0 + (tid + kThreadsT * j) / PR;
            const bool in = ncol && t < g.T;
            mq[MASK ? j : 0] = rsrc_load_bf16x8(mr_, in ? static_cast<unsigned>((t * g.ldm + tail_n) * 2) : kRsrcOob, 0u);
            rq[RES ? j : 0] = rsrc_load_bf16x8(rr_, (has_res && in) ? static_cast<unsigned>((t * g.ldr + tail_n) * 2) : kRsrcOob, 0u);
        }
    };
    auto drain = [&]() __attribute__((always_inline)) {         // (behind the barrier that follows park) the tile leaves row by row:
        // bias, residual, ReLU, dropout, ONE rounding, 16-byte stores
#pragma unroll
        for (int j0 = 0; j0 < NP; j0 += NPB) {
            f32x4 cq[NPB][2];
#pragma unroll
            for (int u = 0; u < NPB; ++u) {
                const float *cr = Cs + ((tid + kThreadsT * (j0 + u)) / PR) * (BN + kCPad) + pc * 8;
                cq[u][0] = *reinterpret_cast<const f32x4 *>(cr);
                cq[u][1] = *reinterpret_cast<const f32x4 *>(cr + 4);
            }
#pragma unroll
            for (int u = 0; u < NPB; ++u) {
                const int j = j0 + u;
                float o[8] = {cq[u][0].x, cq[u][0].y, cq[u][0].z, cq[u][0].w, cq[u][1].x, cq[u][1].y, cq[u][1].z, cq[u][1].w};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float f = o[i] + bv[i];
                    if (RES) f += static_cast<float>(rq[j][i]);
                    if (MASK) f = static_cast<float>(mq[j][i]) <= 0.f ? 0.f : f;       // (threshold_backward's test: a NaN in the mask lets the gradient pass)
                    f = f < lo ? lo : f;                         // ReLU (NaN passes through, as clamp_min does)
                    if (TAIL == 2 && g.thresh) {
                        const uint64_t t = static_cast<uint64_t>(tail_m0 + (tid + kThreadsT * j) / PR);
                        f = ln_hash(sd, t * static_cast<uint64_t>(g.N) + static_cast<uint64_t>(tail_n + i)) >= g.thresh ? f * g.keep_scale : 0.f;
                    }
                    o[i] = f;
                }
                if (out32) {
                    f32x4 o0, o1;
                    o0.x = o[0]; o0.y = o[1]; o0.z = o[2]; o0.w = o[3]; o1.x = o[4]; o1.y = o[5]; o1.z = o[6]; o1.w = o[7];
                    const unsigned off = yoff[j] == kRsrcOob ? kRsrcOob : yoff[j] * 4u;
                    rsrc_store_f32x4(yr, o0, off, 0u);
                    rsrc_store_f32x4(yr, o1, off, 16u);
                } else {
                    bf16x8 ob;
#pragma unroll
                    for (int i = 0; i < 8; ++i) ob[i] = static_cast<__bf16>(o[i]);
                    rsrc_store_bf16x8(yr, ob, yoff[j] == kRsrcOob ? kRsrcOob : yoff[j] * 2u, 0u);
                }
            }
        }
    };

    // ---- the slab sequence.  Compute cursor (cv, ck), fetch cursor (fv, fk) up to 1 + PF slabs ahead; slab s of the sequence lives
    // in LDS buffer s & 1 and, before that, in register set (s - 1) % PF.
    int cv = next_tile(static_cast<int>(blockIdx.x));
    if (cv >= vtiles) return;
    int ck = 0, fv = cv, fk = 0;
    auto advance = [&]() __attribute__((always_inline)) {        // the fetch cursor moves one slab on
        if (++fk == KT) {
            fk = 0;
            fv = next_tile(fv + G);
            if (fv < vtiles) aim(fv);
        }
    };
    // prologue: slab 0 goes through the LAST register set, so that slabs 1 .. PF - 1 (sets 0 .. PF - 2, where the loop expects them)
    // are requested before the first wait -- with slab 0 in set 0 the kernel's first two round trips were one after the other
    TG_MARK(1);
    aim(fv);
    fetch(0, xr[PF - 1], wr[PF - 1]);
    advance();
#pragma unroll
    for (int p = 0; p < PF - 1; ++p)
        if (fv < vtiles) { fetch(fk, xr[p], wr[p]); advance(); }
    TG_MARK(2);
    deposit(0, xr[PF - 1], wr[PF - 1]);
    TG_MARK(3);
    if (fv < vtiles) { fetch(fk, xr[PF - 1], wr[PF - 1]); advance(); }
    clear();
    __syncthreads();
    TG_MARK(4);
    for (int s = 0;; s += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const bool last = ck == KT - 1;                      // (uniform) this tile's last slab
            if (!last) {
                deposit((s + p + 1) & 1, xr[p], wr[p]);          // slab s + p + 1 (the barrier behind the previous slab freed that buffer)
                if (fv < vtiles) { fetch(fk, xr[p], wr[p]); advance(); }
            }
            TG_MARK(5 + 3 * (s + p));
            products((s + p) & 1);
            TG_MARK(6 + 3 * (s + p));
            __syncthreads();
            TG_MARK(7 + 3 * (s + p));
            ++ck;
            if (last) {                                          // every wave is done with the slab buffers: the tile is parked there
                aim_tail(cv);
                park();
                TG_MARK(40);
                if (MASK) {
                    __builtin_amdgcn_sched_barrier(0);           // (the requests stay behind the accumulators' last reads: their registers are the room)
                    aim_mask();
                }
                __syncthreads();
                TG_MARK(41);
                drain();                                         // the tile leaves
                TG_MARK(42);
                cv = next_tile(cv + G);
                if (cv >= vtiles) return;
                __syncthreads();                                 // the buffers are free again
                deposit((s + p + 1) & 1, xr[p], wr[p]);          // the next tile's first slab (in flight since this tile's last slabs)
                if (fv < vtiles) { fetch(fk, xr[p], wr[p]); advance(); }
                clear();
                ck = 0;
                __syncthreads();
            }
        }
    }
}

template <int BM, int BN, bool NN, int PF, int TAIL, int NTH = 256>
hipError_t launch_tile(TgemmArgs g, hipStream_t st)
{
    constexpr size_t lds = tgemm_lds<BM, BN>();
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    constexpr int kThreadsT = NTH;
    auto kern = tgemm_kernel<BM, BN, NN, PF, TAIL, NTH>;
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
    g.tiles_m = static_cast<int>(tiles_m);
    g.ny = (g.N + BN - 1) / BN;
    g.gx = static_cast<int>((tiles_m + 7) / 8 * 8);              // whole rounds over the XCDs (dead row tiles are skipped)
    const int64_t vtiles = static_cast<int64_t>(g.gx) * g.ny;
    // persistent: as many workgroups as the chip holds at once (LDS: 160 KB per CU), a multiple of 8
    int per_cu = static_cast<int>((160 * 1024) / lds);
    per_cu = per_cu > 4 ? 4 : per_cu;
    int64_t grid = static_cast<int64_t>(256) * per_cu;
    { const int f = tune_int("tgemm_grid", 0); if (f >= 8 && f % 8 == 0) grid = f; }       // tests: few workgroups, many tiles each
    if (grid > vtiles) grid = vtiles;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(kThreadsT), lds, st, g);
    return hipGetLastError();
}

template <bool NN, int PF, int TAIL>
hipError_t launch_any(const TgemmArgs &g, int bm, int bn, hipStream_t st)
{
    if (bm == 128 && bn == 128) {
        // the plain forward product takes the big tile with EIGHT waves (2 x 4) and one register set: 120 registers, twice the waves
        // in flight per CU at the same LDS -- 24.3 -> 22.4 us at the encoder shape, 34.7 -> 32.1 at the packed projection
        // (profiles/r05o_gemmbench_eight_waves.json); every other form would spill at the 128-register ceiling that costs.
        // MDETR_TUNE="tgemm_waves=4" (tests) restores four waves.
        if constexpr (!NN && PF == 1 && TAIL == 0) {
            if (tune_int("tgemm_waves", 8) != 4) return launch_tile<128, 128, NN, PF, TAIL, 512>(g, st);
        }
        return launch_tile<128, 128, NN, PF, TAIL>(g, st);
    }
    if (bm == 128) return launch_tile<128, 64, NN, PF, TAIL>(g, st);
    if (bn == 128) return launch_tile<64, 128, NN, PF, TAIL>(g, st);
    return launch_tile<64, 64, NN, PF, TAIL>(g, st);
}

template <bool NN, int PF>
hipError_t launch_tail(const TgemmArgs &g, int bm, int bn, hipStream_t st)
{
    if ((g.flags & (kTgemmBiasF32 | kTgemmOutF32)) || g.thresh) return launch_any<NN, PF, 2>(g, bm, bn, st);
    if constexpr (NN) {
        if (g.mask) return launch_any<NN, PF, 3>(g, bm, bn, st);
    }
    return g.res ? launch_any<NN, PF, 1>(g, bm, bn, st) : launch_any<NN, PF, 0>(g, bm, bn, st);
}

}  // namespace

bool tgemm_supported(const TgemmProblem &p)
{
    const auto al = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool nn = (p.flags & kTgemmNN) != 0;
    return p.T > 0 && p.T < (1ll << 31) - 256 && p.N > 0 && p.N % 8 == 0 && p.K > 0 && p.K % 8 == 0 && p.a && p.w && p.y && al(p.a) && al(p.w) &&
           al(p.y) && p.lda % 8 == 0 && p.lda >= p.K && p.ldw % 8 == 0 && p.ldw >= (nn ? p.N : p.K) &&
           p.ldy % ((p.flags & kTgemmOutF32) ? 4 : 8) == 0 && p.ldy >= p.N &&
           (!p.res || (al(p.res) && p.ldr % 8 == 0 && p.ldr >= p.N)) && (!p.bias || al(p.bias)) && p.dropout_p >= 0.f && p.dropout_p < 1.f &&
           // the masked tail: input gradients only (NN), bf16 in and out, no bias / ReLU / dropout of its own
           (!p.mask || (nn && al(p.mask) && p.ldm % 8 == 0 && p.ldm >= p.N && p.T * p.ldm < (1ll << 30) && !p.bias && p.dropout_p == 0.f &&
                        !(p.flags & (kTgemmRelu | kTgemmBiasF32 | kTgemmOutF32)))) &&
           // buffer-resource addressing: every tensor below 2^31 bytes
           p.T * p.lda < (1ll << 30) && p.T * p.ldy < (1ll << 29) && p.T * p.ldr < (1ll << 30) && static_cast<int64_t>(nn ? p.K : p.N) * p.ldw < (1ll << 30);
}

hipError_t tgemm_launch(const TgemmProblem &p, hipStream_t st)
{
    TgemmArgs g;
    g.a = static_cast<const __bf16 *>(p.a); g.w = static_cast<const __bf16 *>(p.w); g.res = static_cast<const __bf16 *>(p.res);
    g.mask = static_cast<const __bf16 *>(p.mask); g.ldm = p.ldm;
    g.bias = p.bias; g.y = p.y;
    g.T = p.T; g.lda = p.lda; g.ldw = p.ldw; g.ldr = p.ldr; g.ldy = p.ldy;
    g.N = p.N; g.K = p.K; g.gx = g.ny = 0; g.flags = p.flags;
    g.thresh = p.dropout_p > 0.f ? ln_threshold(p.dropout_p) : 0u;
    g.keep_scale = p.dropout_p > 0.f ? 1.f / (1.f - p.dropout_p) : 1.f;
    g.seed = p.seed; g.seed_dev = p.seed_dev;
    // Tile choice (profiles/r05e_gemmbench.json: every product of the step x every tile): 128 x 128 wherever it still gives the
    // chip ~1.5 workgroups per CU, 128 x 64 when only that does, 64 x 64 for narrow outputs, the decoder's few thousand rows and
    // the everything-tail (whose 128 x 128 form runs out of registers)
    const bool generic_tail = (p.flags & (kTgemmBiasF32 | kTgemmOutF32)) != 0 || p.dropout_p > 0.f;
    const auto wgs = [&](int m, int n) { return ((p.T + m - 1) / m) * ((p.N + n - 1) / n); };
    int bm = 64, bn = 64;
    if (p.N > 64 && !generic_tail && !(p.T < 8192 && p.K <= 256)) {
        if (wgs(128, 128) >= 400) { bm = 128; bn = 128; }
        else if (wgs(128, 64) >= 400) { bm = 128; bn = 64; }
    }
    char tune_buf[16];
    if (const char *ev = tune_str("tgemm_tile", tune_buf, sizeof(tune_buf))) {           // tests: "128x64"
        int m = 0, n = 0;
        if (sscanf(ev, "%dx%d", &m, &n) == 2 && (m == 64 || m == 128) && (n == 64 || n == 128)) { bm = m; bn = n; }
    }
    int pf = 2;
    if (bm == 128 && bn == 128 && (p.res || p.mask || (p.flags & (kTgemmBiasF32 | kTgemmOutF32)) || p.dropout_p > 0.f)) pf = 1;       // (the big tile's tails: registers)
    if (bm == 128 && bn == 128 && !(p.flags & kTgemmNN) && !p.res && !generic_tail) pf = 1;                                  // (... and its eight-wave plain form)
    if (const int f = tune_int("tgemm_pf", 0)) pf = f == 1 ? 1 : 2;       // tests: register sets in flight
    ProfileScope prof(10, conv_mflop(p.T, static_cast<int64_t>(p.N) * p.K), st, 2.0 * p.T * p.N * p.K / 1e6,
                      (2.0 * p.T * p.K + ((p.flags & kTgemmOutF32) ? 4.0 : 2.0) * p.T * p.N + (p.res ? 2.0 * p.T * p.N : 0.0) + (p.mask ? 2.0 * p.T * p.N : 0.0) + 2.0 * p.N * p.K) / 1e3);
    if (p.flags & kTgemmNN) return pf == 1 ? launch_tail<true, 1>(g, bm, bn, st) : launch_tail<true, 2>(g, bm, bn, st);
    return pf == 1 ? launch_tail<false, 1>(g, bm, bn, st) : launch_tail<false, 2>(g, bm, bn, st);
}

}  // namespace mdetr
