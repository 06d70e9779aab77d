// monodetr_amd/csrc/twgrad.hip -- dW[N, C] = dY[T, N]^T X[T, C] (+ db[N] = column sums of dY) for the token-wise layers of the
// training iteration: the weight / bias gradients of every 1x1 convolution (torchvision Bottleneck.conv1 / conv3 / downsample behind
// lib/models/monodetr/backbone.py:93-106, the input projections monodetr.py:77-99) and nn.Linear (ops/modules/ms_deform_attn.py:
// 94-102, depthaware_transformer.py:328-353) over tens of thousands of token rows, which the reference leaves to cuBLAS / cuDNN.
//
// The contraction index is the TOKEN, the slow axis of both operands; an MFMA lane needs 8 consecutive contraction values of ONE
// column.  gfx950's transposing LDS read delivers exactly that from a row-major tile: ds_read_b64_tr_b16 hands lane i of a 16-lane
// group the 4 values of column i out of the 4 rows x 16 columns the group's lanes point at (semantics measured on the chip:
// scripts/exp/ds_read_tr16_probe.hip, profiles/r05_ds_read_tr16_probe.txt).  So both operands go into LDS AS THEY LIE IN MEMORY --
// coalesced 16-byte loads, 16-byte LDS stores, no permutes in registers (csrc/conv_wgrad.hip's 1x1 case spent a third of its
// instructions transposing 8 x 8 blocks) -- and a fragment is two transposing reads.
//
// Tiling: a workgroup = 4 waves (2 x 2) owns BN x BC of dW (128 x 128, or 64-wide for narrow operands) and a contiguous chunk of
// the token slabs (TS = 32 tokens); wave tile (BN / 2) x (BC / 2) as 32 x 32 accumulator blocks; a slab's rows are padded by 32
// bf16 (64 bytes), which puts the 4 rows of a transposing read on 4 distinct quarter bank rows.  Register-staged pipeline as in
// tgemm.hip: slab s + 1 is written to LDS behind the barrier that freed its buffer, slabs s + 2 .. s + 1 + PF are in flight.
// Workgroups of one token chunk (the output tiles) are ids 8 apart -- same XCD, adjacent slots -- so the second reader of an
// operand slab finds it in that XCD's L2.  Output: fp32 partials [chunk][N x C (+ N)], summed in a fixed order by colsum.hip
// (deterministic, no atomics).  db rides on the dY fragments already in registers: one more MFMA against a matrix of ones on the
// waves of the first column tile.
// Algorithmic bytes = 2 T (N + C) + 4 N C; flops = 2 T N C.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mdetr_wave.h>

#include "msda.h"       // profile scopes
#include "twgrad.h"
#include "mdetr_tune.h"

namespace mdetr {
namespace {

constexpr int kThreadsW = 256;
constexpr int kRowPad = 32;                // bf16 of padding per slab row (64 bytes)

struct TwgradArgs {
    const __bf16 *x, *dy;
    float *part;
    int64_t T, ldx, ldy, part_stride;
    int C, N, tiles_n, tiles_c, chunks, slabs, slabs_per_chunk, with_db;
};

// fragment of a 32-column block at element `blk` (its first column): lane l -> column l & 31, tokens 8 (l >> 5) .. + 7 of the k-step
// whose first row is `rows`; two transposing reads of 4 tokens each.  Lane s of a 16-lane group points at row s >> 2, columns 4 (s & 3).
__device__ __forceinline__ bf16x8 tr_fragment(const __bf16 *rows, int pitch, int lane)
{
    const int s = lane & 15, grp = lane >> 4;
    const __bf16 *p = rows + (8 * (grp >> 1) + (s >> 2)) * pitch + 16 * (grp & 1) + 4 * (s & 3);
    const bf16x4 lo = lds_read_tr4(p), hi = lds_read_tr4(p + 4 * pitch);
    bf16x8 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = lo[i]; f[4 + i] = hi[i]; }
    return f;
}

// KG = 2: TWO groups of four waves per workgroup, each with the pipeline and the LDS buffers of a four-wave workgroup, on the even /
// odd slabs of the chunk; their accumulators are added through LDS at the end.  Same waves per CU as two workgroups of four -- and
// HALF the fp32 partials to write and to sum again (a partial costs 4 BN BC bytes per workgroup whatever the tile: at [81 600, 256] x
// [81 600, 256] 128 chunks x 256 KB = 33 MB written and read back, against 84 MB of operands).
template <int BN, int BC, int TS, int PF, int KG>
__global__ __launch_bounds__(kThreadsW * KG)
void twgrad_kernel(const TwgradArgs g)
{
    constexpr int PN = BN + kRowPad, PC = BC + kRowPad;          // row pitches of the two slab images
    constexpr int TN = BN / 64, TC = BC / 64;                    // 32 x 32 blocks of a wave along n / c
    constexpr int YCH = TS * BN / 8 / kThreadsW, XCH = TS * BC / 8 / kThreadsW;      // 16-byte pieces per thread and slab
    static_assert(YCH >= 1 && XCH >= 1, "a slab gives every thread at least one piece of each operand");
    MDETR_DYNAMIC_LDS(unsigned char, tw_smem);
    const int kg = KG == 1 ? 0 : wave_uniform(static_cast<int>(threadIdx.x) >> 8);      // this wave's group
    __bf16 *Ys = reinterpret_cast<__bf16 *>(tw_smem) + kg * 2 * TS * (PN + PC);          // [2][TS][PN] of the group
    __bf16 *Xs = Ys + 2 * TS * PN;                               // [2][TS][PC]
    const int tid = threadIdx.x & (kThreadsW - 1), lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int wn = wave & 1, wc = wave >> 1;
    const int tiles = g.tiles_n * g.tiles_c;
    const int id = blockIdx.x, grp = id >> 3;
    const int tile = grp % tiles, chunk = (grp / tiles) * 8 + (id & 7);
    if (chunk >= g.chunks) return;                               // (the chunk count was rounded up to a multiple of 8)
    const int n0 = (tile / g.tiles_c) * BN, c0 = (tile % g.tiles_c) * BC;
    const int s_begin = chunk * g.slabs_per_chunk;
    const int s_end = s_begin + g.slabs_per_chunk < g.slabs ? s_begin + g.slabs_per_chunk : g.slabs;

    // buffer-resource loads: rows beyond T and columns beyond N / C give zeros (they add nothing to the sums)
    const mdetr_rsrc yr = make_rsrc(g.dy, static_cast<unsigned>(((g.T - 1) * g.ldy + g.N) * 2));
    const mdetr_rsrc xr = make_rsrc(g.x, static_cast<unsigned>(((g.T - 1) * g.ldx + g.C) * 2));
    int yrow[YCH], xrow[XCH], ydst[YCH], xdst[XCH];
    unsigned ycol[YCH], xcol[XCH];                               // byte offset of the piece inside its row (kRsrcOob: dead column)
#pragma unroll
    for (int j = 0; j < YCH; ++j) {
        const int c = tid + kThreadsW * j, row = c / (BN / 8), pc = c % (BN / 8);
        yrow[j] = row;
        ycol[j] = n0 + pc * 8 < g.N ? static_cast<unsigned>((n0 + pc * 8) * 2) : kRsrcOob;
        ydst[j] = row * PN + pc * 8;
    }
#pragma unroll
    for (int j = 0; j < XCH; ++j) {
        const int c = tid + kThreadsW * j, row = c / (BC / 8), pc = c % (BC / 8);
        xrow[j] = row;
        xcol[j] = c0 + pc * 8 < g.C ? static_cast<unsigned>((c0 + pc * 8) * 2) : kRsrcOob;
        xdst[j] = row * PC + pc * 8;
    }
    bf16x8 yst[PF][YCH], xst[PF][XCH];
    // (slab index i of this GROUP: the chunk's slab s_begin + KG i + kg; one beyond the chunk -- the groups run the same number of
    // iterations, they share the barriers -- reads zeros)
    auto fetch = [&](int i, bf16x8 (&ys_)[YCH], bf16x8 (&xs_)[XCH]) __attribute__((always_inline)) {
        const int s = s_begin + KG * i + kg;
        const int64_t t0 = static_cast<int64_t>(s) * TS;
        const bool live = s < s_end;
#pragma unroll
        for (int j = 0; j < YCH; ++j) {
            const int64_t t = t0 + yrow[j];
            ys_[j] = rsrc_load_bf16x8(yr, (live && t < g.T && ycol[j] != kRsrcOob) ? static_cast<unsigned>(t * g.ldy * 2) + ycol[j] : kRsrcOob, 0u);
        }
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            const int64_t t = t0 + xrow[j];
            xs_[j] = rsrc_load_bf16x8(xr, (live && t < g.T && xcol[j] != kRsrcOob) ? static_cast<unsigned>(t * g.ldx * 2) + xcol[j] : kRsrcOob, 0u);
        }
    };
    auto deposit = [&](int buf, const bf16x8 (&ys_)[YCH], const bf16x8 (&xs_)[XCH]) __attribute__((always_inline)) {
        __bf16 *yb = Ys + buf * TS * PN, *xb = Xs + buf * TS * PC;
#pragma unroll
        for (int j = 0; j < YCH; ++j) *reinterpret_cast<bf16x8 *>(yb + ydst[j]) = ys_[j];
#pragma unroll
        for (int j = 0; j < XCH; ++j) *reinterpret_cast<bf16x8 *>(xb + xdst[j]) = xs_[j];
    };

    f32x16 acc[TN][TC], accb[TN];
#pragma unroll
    for (int a_ = 0; a_ < TN; ++a_) {
#pragma unroll
        for (int i = 0; i < 16; ++i) accb[a_][i] = 0.f;
#pragma unroll
        for (int b_ = 0; b_ < TC; ++b_)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a_][b_][i] = 0.f;
    }
    const bool want_db = g.with_db != 0 && c0 == 0 && wc == 0;   // (wave-uniform) the first column tile's first column waves carry db
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = static_cast<__bf16>(1.0f);

    auto products = [&](int buf) __attribute__((always_inline)) {
        const __bf16 *yb = Ys + buf * TS * PN + wn * (BN / 2), *xb = Xs + buf * TS * PC + wc * (BC / 2);
#pragma unroll
        for (int ks = 0; ks < TS / 16; ++ks) {
            bf16x8 yf[TN], xf[TC];
#pragma unroll
            for (int a_ = 0; a_ < TN; ++a_) yf[a_] = tr_fragment(yb + ks * 16 * PN + a_ * 32, PN, lane);
#pragma unroll
            for (int b_ = 0; b_ < TC; ++b_) xf[b_] = tr_fragment(xb + ks * 16 * PC + b_ * 32, PC, lane);
#pragma unroll
            for (int a_ = 0; a_ < TN; ++a_) {
#pragma unroll
                for (int b_ = 0; b_ < TC; ++b_) acc[a_][b_] = mfma_bf16(yf[a_], xf[b_], acc[a_][b_]);     // D[n][c] += dY^T[n][8 t] X[8 t][c]
                if (want_db) accb[a_] = mfma_bf16(yf[a_], ones, accb[a_]);                             // every column: the k-step's sum over t
            }
        }
    };

    // ---- slab pipeline (tgemm.hip's): slab s + 1 is written to LDS after the barrier that freed its buffer, PF more are in flight
    const int ns = (s_end - s_begin + KG - 1) / KG;              // iterations of a group
    if (ns > 0) {
        // (slab 0 through the last register set: slabs 1 .. PF - 1 are requested before the first wait, as in tgemm.hip)
        fetch(0, yst[PF - 1], xst[PF - 1]);
#pragma unroll
        for (int p = 0; p < PF - 1; ++p)
            if (1 + p < ns) fetch(1 + p, yst[p], xst[p]);
        deposit(0, yst[PF - 1], xst[PF - 1]);
        if (PF < ns) fetch(PF, yst[PF - 1], xst[PF - 1]);
        __syncthreads();
        for (int k = 0; k < ns; k += PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int s = k + p;
                if (s < ns) {                                    // (uniform)
                    if (s + 1 < ns) deposit((s + 1) & 1, yst[p], xst[p]);
                    if (s + 1 + PF < ns) fetch(s + 1 + PF, yst[p], xst[p]);
                    products(s & 1);
                    __syncthreads();
                }
            }
        }
    }
    if (KG == 2) {
        // the second group's accumulators through LDS (register-major: lane-contiguous 256-byte rows, 64 KB for 128 x 128), added by the
        // first group's wave of the same quadrant: one partial per workgroup
        float *red = reinterpret_cast<float *>(tw_smem);
        constexpr int kPerWave = TN * TC * 16 * 64 + TN * 32;     // (the bias sums are the same in all 32 columns: one lane per half stores them)
        float *mine = red + wave * kPerWave;
        if (kg == 1) {
#pragma unroll
            for (int a_ = 0; a_ < TN; ++a_) {
#pragma unroll
                for (int b_ = 0; b_ < TC; ++b_)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[((a_ * TC + b_) * 16 + r) * 64 + lane] = acc[a_][b_][r];
                if (l31 == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mine[TN * TC * 16 * 64 + (a_ * 16 + r) * 2 + half] = accb[a_][r];
                }
            }
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int a_ = 0; a_ < TN; ++a_) {
#pragma unroll
            for (int b_ = 0; b_ < TC; ++b_)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a_][b_][r] += mine[((a_ * TC + b_) * 16 + r) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 16; ++r) accb[a_][r] += mine[TN * TC * 16 * 64 + (a_ * 16 + r) * 2 + half];
        }
    }

    // ---- this chunk's partial: acc[a][b] register r of lane l = D[n = 32 a + (r & 3) + 8 (r >> 2) + 4 (l >> 5)][c = 32 b + (l & 31)]
    float *pp = g.part + static_cast<int64_t>(chunk) * g.part_stride;
#pragma unroll
    for (int a_ = 0; a_ < TN; ++a_) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * (BN / 2) + a_ * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (n >= g.N) continue;
            if (want_db && l31 == 0) pp[static_cast<int64_t>(g.N) * g.C + n] = accb[a_][r];
#pragma unroll
            for (int b_ = 0; b_ < TC; ++b_) {
                const int c = c0 + wc * (BC / 2) + b_ * 32 + l31;
                if (c < g.C) pp[static_cast<int64_t>(n) * g.C + c] = acc[a_][b_][r];
            }
        }
    }
}

struct TwgradPlan {
    int bn, bc, tiles_n, tiles_c, chunks, slabs, slabs_per_chunk, kg;
};

constexpr int kSlab = 32;

TwgradPlan plan(int64_t T, int C, int N)
{
    TwgradPlan p;
    p.bn = N <= 64 ? 64 : 128;
    p.bc = C <= 64 ? 64 : 128;
    p.tiles_n = (N + p.bn - 1) / p.bn;
    p.tiles_c = (C + p.bc - 1) / p.bc;
    p.slabs = static_cast<int>((T + kSlab - 1) / kSlab);
    // two workgroups per CU, every chunk with at least four slabs: more chunks mean more fp32 partials to write and sum again
    // (256 / 512 / 1024 workgroups at [81 600, 256] x [81 600, 256]: 39.1 / 32.6 / 46.9 us with the chunk sum, profiles/r05h_wgradbench.json)
    // Two wave groups per workgroup (see the kernel) for the full tile on long token axes: half the workgroups, half the partials
    // (profiles/r06_wgradbench_kg.json).
    p.kg = (p.bn == 128 && p.bc == 128 && p.slabs >= 256) ? 2 : 1;
    { const int f = tune_int("twgrad_kg", 0); if (f == 1 || (f == 2 && p.bn == 128 && p.bc == 128)) p.kg = f; }      // tests / A-B runs
    int target = 512 / p.kg;
    { const int f = tune_int("twgrad_wgs", 0); if (f >= 64 && f <= 8192) target = f; }       // tests
    int chunks = target / (p.tiles_n * p.tiles_c);
    // chunk c runs on XCD c % 8 (all its tiles: the second reader of a slab finds it in that L2), so the chunks come in whole
    // eights -- 42 chunks of 6 tiles put 36 workgroups on the 32 CUs of two XCDs and 30 on the others: a second round on two
    if (chunks >= 8) chunks = chunks / 8 * 8;
    if (chunks > p.slabs / (4 * p.kg)) chunks = p.slabs / (4 * p.kg);
    if (chunks < 1) chunks = 1;
    p.slabs_per_chunk = (p.slabs + chunks - 1) / chunks;
    p.chunks = (p.slabs + p.slabs_per_chunk - 1) / p.slabs_per_chunk;          // every chunk has at least one slab
    return p;
}

template <int BN, int BC, int KG>
hipError_t launch_tile(TwgradArgs g, hipStream_t st)
{
    constexpr int TS = kSlab, PF = 2;
    constexpr size_t slabs_b = static_cast<size_t>(KG) * 2 * TS * ((BN + kRowPad) + (BC + kRowPad)) * 2;
    constexpr size_t red_b = KG == 2 ? static_cast<size_t>(4) * ((BN / 64) * (BC / 64) * 16 * 64 + (BN / 64) * 32) * 4 : 0;
    constexpr size_t lds = slabs_b > red_b ? slabs_b : red_b;
    auto kern = twgrad_kernel<BN, BC, TS, PF, KG>;
    static bool attr_set[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = -1;
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    const int64_t grid = static_cast<int64_t>((g.chunks + 7) / 8 * 8) * g.tiles_n * g.tiles_c;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(kThreadsW * KG), lds, st, g);
    return hipGetLastError();
}

}  // namespace

bool twgrad_supported(int64_t T, int C, int N, int64_t ldx, int64_t ldy, const void *x, const void *dy)
{
    const auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return T > 0 && C > 0 && N > 0 && C % 8 == 0 && N % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= N && al(x) && al(dy) &&
           T * ldx < (1ll << 30) && T * ldy < (1ll << 30) && static_cast<int64_t>(N) * C < (1ll << 28) && T < (1ll << 31) - 64;
}

int twgrad_chunks(int64_t T, int C, int N) { return plan(T, C, N).chunks; }

hipError_t twgrad_launch(const void *x, const void *dy, float *part, int64_t T, int C, int N, int64_t ldx, int64_t ldy, bool with_db,
                         hipStream_t st)
{
    const TwgradPlan p = plan(T, C, N);
    TwgradArgs g;
    g.x = static_cast<const __bf16 *>(x); g.dy = static_cast<const __bf16 *>(dy); g.part = part;
    g.T = T; g.ldx = ldx; g.ldy = ldy; g.part_stride = static_cast<int64_t>(N) * C + (with_db ? N : 0);
    g.C = C; g.N = N; g.tiles_n = p.tiles_n; g.tiles_c = p.tiles_c; g.chunks = p.chunks; g.slabs = p.slabs; g.slabs_per_chunk = p.slabs_per_chunk;
    g.with_db = with_db ? 1 : 0;
    ProfileScope prof(11, conv_mflop(T, static_cast<int64_t>(C) * N), st, 2.0 * T * N * C / 1e6,
                      (2.0 * T * (N + C) + 4.0 * p.chunks * (static_cast<double>(N) * C + (with_db ? N : 0))) / 1e3);       // (+ the fp32 partials it writes)
    if (p.bn == 128 && p.bc == 128) return p.kg == 2 ? launch_tile<128, 128, 2>(g, st) : launch_tile<128, 128, 1>(g, st);
    if (p.bn == 128) return launch_tile<128, 64, 1>(g, st);
    if (p.bc == 128) return launch_tile<64, 128, 1>(g, st);
    return launch_tile<64, 64, 1>(g, st);
}

}  // namespace mdetr
