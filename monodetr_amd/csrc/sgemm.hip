// monodetr_amd/csrc/sgemm.hip -- grouped fp32 products on the f32-input matrix instruction (v_mfma_f32_32x32x2_f32).
//
// The prediction heads of MonoDETR (lib/models/monodetr/monodetr.py:222-262: class / box / size / angle / depth MLPs on the
// [B x 550, 256] output of every decoder level) stay in fp32 behind the bf16 body.  They are 18 small products per level and direction
// -- [4 400, 256] x [256, 256], x [256, 6], x [256, 3] ... -- which the framework hands to the library one launch at a time (60
// launches, 0.96 ms of the round-5 step, most of it fill and drain).  This file runs them as GROUPS: one launch covers every product
// of a dependency level (the five first layers; the second layers; ...), a workgroup looks its tile up in the group's table.
//
// Arithmetic: `mfma_f32` is exact fp32 (the k-ordered fmaf chain), at the fp32 vector rate -- 1/16 of the bf16 instruction, which is
// still 155 TFLOP/s; the heads are 27 GFLOP per step.  No operand is rounded to a narrower type anywhere (bf16 OPERANDS -- the
// decoder output of the bf16 body -- are widened on load, which is exact).
//
// Layout of a product C[M, N] = sum_t A_t op(B_t):
//   NT / NN: a 256-thread workgroup owns a 64 x 64 tile, its four waves a 32 x 32 quadrant each; the contraction runs in slabs of 32
//            through LDS (double-buffered; the next slab's global loads fly during the current slab's 16 matrix instructions).  An
//            operand whose contraction index is contiguous in memory (A always; B in NT) is staged [row][32 + 4] and read with one
//            ds_read_b128 per four instructions (lane (i, h) takes k = 8 q + 4 h + 0..3: both operands use the same assignment, so the
//            products pair up correctly whatever the order); NN's B is staged [k][64 + 4] and read with ds_read_b32.
//   TN:      C[M, N] = A[K, M]^T B[K, N] contracts over the ROW index of both operands (K = 4 400 tokens): the instruction's operand
//            layout (32 consecutive output rows per half wave, one contraction index per half) is how such operands lie in memory,
//            so lanes load straight from global memory, no LDS.  Lane (i, h) loads the column PAIRS A[t + h][m0 + 2 i, + 1] and
//            B[t + h][n0 + 2 i, + 1] (one 8-byte load each) and issues four instructions on them -- even / odd columns of either
//            operand, a 64 x 64 tile per wave in four accumulators whose rows / columns interleave.  (One value per load and
//            instruction, the first version, was bound by the L2 at 106 us for a level's weight gradients.)  A 1024-thread workgroup
//            owns the tile; its sixteen waves take a sixteenth of the rows each and are summed pairwise through LDS in a fixed tree
//            (deterministic); the column sums of A (the bias gradient) ride on the A operand's registers.
// Ragged shapes are guarded per element; vector loads are used where the stride and base pointer allow them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mdetr_wave.h>

#include "msda.h"        // ProfileScope
#include "sgemm.h"

namespace mdetr {
namespace {

constexpr int kTile = 64, kSlab = 32, kPitchK = kSlab + 4, kPitchN = kTile + 4, kTnTile = 64, kTnWaves = 4, kTnMaxSplit = 16;
constexpr int kMaxTerms = 24;                        // terms of all problems of a group, pooled
constexpr int kFlagCBf16 = 1, kFlagResBf16 = 2;

struct Term {
    const void *a, *b;
    int64_t lda, ldb;
    int K, a_bf16, b_bf16, pad_;
};
struct Prob {
    void *c;
    const float *bias;
    float *colsum;
    const float *mask;
    const void *res;
    int64_t ldc, ldm, ldr;
    int M, N, relu_cols, term0, nterm, tile0, tiles_n, flags;
    int foff, pad_;                    // TN with a split contraction: where the problem's M x N result (+ its M column sums) starts in a partial
};
struct Args {
    Term t[kMaxTerms];
    Prob p[MDETR_SGEMM_MAX_PROBLEMS];
    int nprob;
    int split, flat;                   // TN: the contraction in `split` parts (grid.y), each writing a partial of `flat` floats to ws
    float *ws;
};
// The group's table is a kernel ARGUMENT (2 KB by value: a captured graph replays it, no descriptor buffer to keep alive); kernels
// copy it into LDS first thing (stage_args) and never touch the argument block again.

__device__ __forceinline__ float bf16_bits_to_float(unsigned short v) { return __uint_as_float(static_cast<unsigned>(v) << 16); }

// Loads never sit behind a per-lane branch (a branchy guard makes the compiler wait for every load before it issues the next: the
// first version of this file ran its 32 loads per step one memory latency after another).  Element types and "may this slab be read
// with vector loads" are UNIFORM decisions taken once per slab; a lane whose element does not exist loads element 0 of the operand
// instead (always mapped) and selects zero.
template <bool BF16> __device__ __forceinline__ float ld1(const void *base, int64_t off)
{
    if (BF16) return bf16_bits_to_float(as_global<unsigned short>(base)[off]);
    return as_global<float>(base)[off];
}
template <bool BF16> __device__ __forceinline__ float ld1_if(const void *base, int64_t off, bool ok)
{
    const float v = ld1<BF16>(base, ok ? off : 0);
    return ok ? v : 0.f;
}
// The NT / NN staging loads are split in two: `raw4` only ISSUES the loads (no instruction consumes a loaded value, so nothing waits
// for memory before the slab's matrix instructions); `decode4` -- widening, zeroing of missing elements -- runs when the registers are
// written to LDS, after those instructions.  (With the conversion next to the load every load was followed by s_waitcnt vmcnt(0):
// four serial memory latencies per slab, 3 us per slab against 0.46 us of matrix instructions.)
// `full`: (uniform) every lane's four elements exist along the contiguous axis and vector loads are aligned; valid = 0 or 4 then.
__device__ __forceinline__ u32x4 raw4(const void *base, int bf16, int64_t off, int valid, bool full)
{
    u32x4 r;
    if (full) {
        const int64_t o = valid > 0 ? off : 0;
        if (bf16) {                                                // (z, w stay unset: decode4 does not read them in this form -- setting them
            const u32x2 v = *(const MDETR_GLOBAL u32x2 *)(as_global<unsigned short>(base) + o);   // would cost a register copy of the
            r.x = v.x; r.y = v.y;                                  // loaded pair, i.e. a wait for the load right here)
        } else {
            r = *(const MDETR_GLOBAL u32x4 *)(as_global<float>(base) + o);
        }
    } else if (bf16) {
        const MDETR_GLOBAL unsigned short *p = as_global<unsigned short>(base);
        r.x = p[valid > 0 ? off : 0]; r.y = p[valid > 1 ? off + 1 : 0]; r.z = p[valid > 2 ? off + 2 : 0]; r.w = p[valid > 3 ? off + 3 : 0];
    } else {
        const MDETR_GLOBAL unsigned *p = as_global<unsigned>(base);
        r.x = p[valid > 0 ? off : 0]; r.y = p[valid > 1 ? off + 1 : 0]; r.z = p[valid > 2 ? off + 2 : 0]; r.w = p[valid > 3 ? off + 3 : 0];
    }
    return r;
}
__device__ __forceinline__ f32x4 decode4(u32x4 r, int bf16, int valid, bool full)
{
    f32x4 f;
    if (!bf16) {
        f.x = __uint_as_float(r.x); f.y = __uint_as_float(r.y); f.z = __uint_as_float(r.z); f.w = __uint_as_float(r.w);
    } else if (full) {
        f.x = __uint_as_float(r.x << 16); f.y = __uint_as_float(r.x & 0xffff0000u);
        f.z = __uint_as_float(r.y << 16); f.w = __uint_as_float(r.y & 0xffff0000u);
    } else {
        f.x = __uint_as_float(r.x << 16); f.y = __uint_as_float(r.y << 16); f.z = __uint_as_float(r.z << 16); f.w = __uint_as_float(r.w << 16);
    }
    if (valid < 1) f.x = 0.f;
    if (valid < 2) f.y = 0.f;
    if (valid < 3) f.z = 0.f;
    if (valid < 4) f.w = 0.f;
    return f;
}

__device__ __forceinline__ float load1(const void *base, int bf16, int64_t off)
{
    return bf16 ? ld1<true>(base, off) : ld1<false>(base, off);
}

// can rows of this operand be read with one 8- / 16-byte load per four elements?  (base and row stride aligned; the column offsets the
// kernels use are multiples of four)
__device__ __forceinline__ bool vec_ok(const void *base, int64_t ld, int bf16)
{
    const uintptr_t mask = bf16 ? 7u : 15u;
    return (reinterpret_cast<uintptr_t>(base) & mask) == 0 && (ld & 3) == 0;
}

__device__ __forceinline__ int clamp04(int v) { return v < 0 ? 0 : (v > 4 ? 4 : v); }

// The whole argument block, copied into LDS by all threads in ONE round of independent loads (ends with a barrier), and this
// workgroup's problem looked up there.  Measured: a chain of dependent reads of the argument block (problem count -> tile table ->
// problem -> its terms) cost ~12 us per launch before any arithmetic -- a one-slab product took 12.3 us.
__device__ __forceinline__ const Prob &stage_args(const Args &a, Args *lds, int tile)
{
    const unsigned *src = reinterpret_cast<const unsigned *>(&a);
    unsigned *dst = reinterpret_cast<unsigned *>(lds);
    for (unsigned i = threadIdx.x; i < sizeof(Args) / 4; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    int pi = 0;
    const int nprob = wave_uniform(lds->nprob);
    for (int i = 1; i < nprob; ++i)
        if (tile >= wave_uniform(lds->p[i].tile0)) pi = i;
    return lds->p[pi];
}

// a staged term with every field in scalar registers
__device__ __forceinline__ Term uniform_term(const Term &t)
{
    Term u;
    u.a = reinterpret_cast<const void *>(wave_uniform64(reinterpret_cast<int64_t>(t.a)));
    u.b = reinterpret_cast<const void *>(wave_uniform64(reinterpret_cast<int64_t>(t.b)));
    u.lda = wave_uniform64(t.lda); u.ldb = wave_uniform64(t.ldb);
    u.K = wave_uniform(t.K); u.a_bf16 = wave_uniform(t.a_bf16); u.b_bf16 = wave_uniform(t.b_bf16); u.pad_ = 0;
    return u;
}

// The tail of one 32 x 32 accumulator: rows row0 + rmul * (the instruction's row of register r), one column per lane.  Loads first
// (the lane's bias value once; the 16 mask / res values back to back from clamped addresses), arithmetic, then stores under the
// lane's bounds -- no per-element branch around a load (the first version walked its 16 elements one memory latency at a time:
// 13 us of a 15 us launch).
__device__ __forceinline__ void epilogue_tile(const Prob &P, int row0, int rmul, int col, int lh, const f32x16 &acc)
{
    const int M = P.M, N = P.N;
    const bool colok = col < N;
    const int64_t ldc = P.ldc;
    float b = 0.f;
    if (P.bias) b = as_global<float>(P.bias)[colok ? col : 0];
    float rs[16], mk[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { rs[r] = 0.f; mk[r] = 1.f; }
    if (P.res) {
        const int64_t ldr = P.ldr;
        const int rbf = P.flags & kFlagResBf16;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = row0 + rmul * ((r & 3) + 8 * (r >> 2) + 4 * lh);
            const int64_t o = (gm < M && colok) ? static_cast<int64_t>(gm) * ldr + col : 0;
            rs[r] = rbf ? ld1<true>(P.res, o) : ld1<false>(P.res, o);
        }
    }
    if (P.mask) {
        const int64_t ldm = P.ldm;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = row0 + rmul * ((r & 3) + 8 * (r >> 2) + 4 * lh);
            mk[r] = as_global<float>(P.mask)[(gm < M && colok) ? static_cast<int64_t>(gm) * ldm + col : 0];
        }
    }
    const bool relu = col < P.relu_cols, cbf = (P.flags & kFlagCBf16) != 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = row0 + rmul * ((r & 3) + 8 * (r >> 2) + 4 * lh);
        float v = acc[r] + b + rs[r];
        if (relu) v = fmaxf(v, 0.f);
        if (!(mk[r] > 0.f)) v = 0.f;
        if (gm < M && colok) {
            const int64_t o = static_cast<int64_t>(gm) * ldc + col;
            if (cbf) as_global_rw<__bf16>(P.c)[o] = static_cast<__bf16>(v);
            else as_global_rw<float>(P.c)[o] = v;
        }
    }
}

// ---- NT / NN -----------------------------------------------------------------------------------------------------------------
// The 16 matrix instructions of one staged slab (this wave's 32 x 32 quadrant)
template <bool NN>
__device__ __forceinline__ void slab_products(const float *As, const float *Bs, int wm, int wn, int li, int lh, f32x16 &acc)
{
    const float *ap = As + (wm * 32 + li) * kPitchK + 4 * lh;
    if (NN) {
        const float *bp = Bs + (4 * lh) * kPitchN + wn * 32 + li;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 8 * q);
            acc = mfma_f32(av.x, bp[(8 * q + 0) * kPitchN], acc);
            acc = mfma_f32(av.y, bp[(8 * q + 1) * kPitchN], acc);
            acc = mfma_f32(av.z, bp[(8 * q + 2) * kPitchN], acc);
            acc = mfma_f32(av.w, bp[(8 * q + 3) * kPitchN], acc);
        }
    } else {
        const float *bp = Bs + (wn * 32 + li) * kPitchK + 4 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 8 * q);
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(bp + 8 * q);
            acc = mfma_f32(av.x, bv.x, acc);
            acc = mfma_f32(av.y, bv.y, acc);
            acc = mfma_f32(av.z, bv.z, acc);
            acc = mfma_f32(av.w, bv.w, acc);
        }
    }
}

struct TileCtx {                       // what a term's loop needs to know about the workgroup (uniform except the staging coordinates)
    int M, N, tm, tn;
    int sr, sc, br, bc;                // staging maps: K-contiguous operand row / column of this thread; NN's B k-row / column
    int wm, wn, li, lh;
};

// One term of a product in the FAST form: the contraction is a multiple of the slab, both operands allow 16-byte (fp32) / 8-byte (bf16
// A) loads, B is fp32, NN's tile lies inside N.  No element is ever missing, so nothing is selected or counted: rows beyond M (or
// beyond N for NT's B) read the operand's last row instead and produce results nobody stores.  Pointers advance by one slab per
// iteration; the loop body is loads -> 16 matrix instructions -> LDS stores -> barrier, without a branch on a mode.
// `parity`: the LDS buffer the term's first slab goes to (the buffers alternate across terms as well).
template <bool NN, bool ABF>
__device__ __forceinline__ int term_fast(const Term &T, const TileCtx &c, float (*As)[kTile * kPitchK], float (*Bs)[NN ? kSlab * kPitchN : kTile * kPitchK],
                                         int parity, f32x16 &acc)
{
    const int slabs = T.K / kSlab;
    const MDETR_GLOBAL unsigned short *pa16[2];
    const MDETR_GLOBAL float *pa32[2], *pb[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row = min(c.tm * kTile + c.sr + 32 * p, c.M - 1);
        if (ABF) pa16[p] = as_global<unsigned short>(T.a) + static_cast<int64_t>(row) * T.lda + c.sc;
        else pa32[p] = as_global<float>(T.a) + static_cast<int64_t>(row) * T.lda + c.sc;
        if (NN) pb[p] = as_global<float>(T.b) + static_cast<int64_t>(c.br + 16 * p) * T.ldb + c.tn * kTile + c.bc;
        else pb[p] = as_global<float>(T.b) + static_cast<int64_t>(min(c.tn * kTile + c.br + 32 * p, c.N - 1)) * T.ldb + c.sc;
    }
    const int64_t bstep = NN ? static_cast<int64_t>(kSlab) * T.ldb : kSlab;
    // THREE slabs in flight in registers.  A dependent global load costs ~1 us here (measured: scripts/exp/sgemm_probe.py, 1.1 us per
    // slab with one slab of prefetch) against 0.46 us for a slab's 16 matrix instructions: the loads of slab s + 3 are issued before
    // the products of slab s, so a load has three slabs of products to arrive in.
    u32x4 ra[3][2], rb[3][2];
    u32x2 ha[3][2];
    auto load = [&](u32x4 (&xa)[2], u32x2 (&xh)[2], u32x4 (&xb)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (ABF) { xh[p] = *(const MDETR_GLOBAL u32x2 *)(pa16[p]); pa16[p] += kSlab; }
            else { xa[p] = *(const MDETR_GLOBAL u32x4 *)(pa32[p]); pa32[p] += kSlab; }
            xb[p] = *(const MDETR_GLOBAL u32x4 *)(pb[p]);
            pb[p] += bstep;
        }
    };
    auto store = [&](int buf, const u32x4 (&xa)[2], const u32x2 (&xh)[2], const u32x4 (&xb)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f32x4 a4;
            if (ABF) {
                a4.x = __uint_as_float(xh[p].x << 16); a4.y = __uint_as_float(xh[p].x & 0xffff0000u);
                a4.z = __uint_as_float(xh[p].y << 16); a4.w = __uint_as_float(xh[p].y & 0xffff0000u);
            } else {
                a4.x = __uint_as_float(xa[p].x); a4.y = __uint_as_float(xa[p].y); a4.z = __uint_as_float(xa[p].z); a4.w = __uint_as_float(xa[p].w);
            }
            f32x4 b4;
            b4.x = __uint_as_float(xb[p].x); b4.y = __uint_as_float(xb[p].y); b4.z = __uint_as_float(xb[p].z); b4.w = __uint_as_float(xb[p].w);
            *reinterpret_cast<f32x4 *>(&As[buf][(c.sr + 32 * p) * kPitchK + c.sc]) = a4;
            if (NN) *reinterpret_cast<f32x4 *>(&Bs[buf][(c.br + 16 * p) * kPitchN + c.bc]) = b4;
            else *reinterpret_cast<f32x4 *>(&Bs[buf][(c.br + 32 * p) * kPitchK + c.bc]) = b4;
        }
    };
    load(ra[0], ha[0], rb[0]);
    if (slabs > 1) load(ra[1], ha[1], rb[1]);
    if (slabs > 2) load(ra[2], ha[2], rb[2]);
    store(parity, ra[0], ha[0], rb[0]);
    __syncthreads();
    // slab s2 sits in LDS buffer (parity + s2) & 1; ring slot s2 % 3 is free (its slab is in LDS), slot (s2 + 1) % 3 goes to LDS next
#define MDETR_SGEMM_STEP(FREE, NEXT)                                                                                   \
    {                                                                                                                  \
        const int buf = (parity + s2) & 1;                                                                             \
        if (s2 + 3 < slabs) load(ra[FREE], ha[FREE], rb[FREE]);                                                        \
        slab_products<NN>(As[buf], Bs[buf], c.wm, c.wn, c.li, c.lh, acc);                                              \
        if (s2 + 1 < slabs) store(buf ^ 1, ra[NEXT], ha[NEXT], rb[NEXT]);                                              \
        __syncthreads();                                                                                               \
        if (++s2 >= slabs) break;                                                                                      \
    }
    for (int s2 = 0;;) {
        MDETR_SGEMM_STEP(0, 1)
        MDETR_SGEMM_STEP(1, 2)
        MDETR_SGEMM_STEP(2, 0)
    }
#undef MDETR_SGEMM_STEP
    return (parity + slabs) & 1;
}

// One term in the GENERAL form: any contraction length, any alignment, either element type for either operand -- every load guarded
// (clamped address + a count of existing elements), decoded when it is written to LDS.
template <bool NN>
__device__ __forceinline__ int term_any(const Term &T, const TileCtx &c, float (*As)[kTile * kPitchK], float (*Bs)[NN ? kSlab * kPitchN : kTile * kPitchK],
                                        int parity, f32x16 &acc)
{
    const int slabs = (T.K + kSlab - 1) / kSlab;
    u32x4 ra[2], rb[2];                                            // the slab in flight, as loaded
    int va[2], vb[2];                                              // ... how many of each lane's four elements exist
    bool la_full = false, lb_full = false;                         // ... and how to read the registers (uniform)
    int k0 = 0;
    auto load = [&]() {
        const bool kfull = k0 + kSlab <= T.K;                      // uniform: the slab lies inside the contraction
        la_full = kfull && vec_ok(T.a, T.lda, T.a_bf16);
        lb_full = NN ? ((c.tn + 1) * kTile <= c.N && vec_ok(T.b, T.ldb, T.b_bf16)) : (kfull && vec_ok(T.b, T.ldb, T.b_bf16));
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = c.tm * kTile + c.sr + 32 * p, k = k0 + c.sc;
            va[p] = row < c.M ? clamp04(T.K - k) : 0;
            ra[p] = raw4(T.a, T.a_bf16, static_cast<int64_t>(row) * T.lda + k, va[p], la_full);
            if (NN) {
                const int kk = k0 + c.br + 16 * p, col = c.tn * kTile + c.bc;
                vb[p] = kk < T.K ? clamp04(c.N - col) : 0;
                rb[p] = raw4(T.b, T.b_bf16, static_cast<int64_t>(kk) * T.ldb + col, vb[p], lb_full);
            } else {
                const int rowb = c.tn * kTile + c.br + 32 * p;
                vb[p] = rowb < c.N ? clamp04(T.K - k) : 0;
                rb[p] = raw4(T.b, T.b_bf16, static_cast<int64_t>(rowb) * T.ldb + k, vb[p], lb_full);
            }
        }
        k0 += kSlab;
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            *reinterpret_cast<f32x4 *>(&As[buf][(c.sr + 32 * p) * kPitchK + c.sc]) = decode4(ra[p], T.a_bf16, va[p], la_full);
            const f32x4 b4 = decode4(rb[p], T.b_bf16, vb[p], lb_full);
            if (NN) *reinterpret_cast<f32x4 *>(&Bs[buf][(c.br + 16 * p) * kPitchN + c.bc]) = b4;
            else *reinterpret_cast<f32x4 *>(&Bs[buf][(c.br + 32 * p) * kPitchK + c.bc]) = b4;
        }
    };
    load();
    store(parity);
    __syncthreads();
    for (int s2 = 0; s2 < slabs; ++s2) {
        const int buf = (parity + s2) & 1;
        if (s2 + 1 < slabs) load();
        slab_products<NN>(As[buf], Bs[buf], c.wm, c.wn, c.li, c.lh, acc);
        if (s2 + 1 < slabs) store(buf ^ 1);
        __syncthreads();
    }
    return (parity + slabs) & 1;
}

template <bool NN>
__global__ __launch_bounds__(256)
void sgemm_kernel(const Args a)
{
    __shared__ __attribute__((aligned(16))) float As[2][kTile * kPitchK];
    __shared__ __attribute__((aligned(16))) float Bs[2][NN ? kSlab * kPitchN : kTile * kPitchK];
    __shared__ __attribute__((aligned(16))) Args largs;
    const int tile = blockIdx.x;
    const Prob &P = stage_args(a, &largs, tile);
    const Term *terms = &largs.t[wave_uniform(P.term0)];
    const int tiles_n = wave_uniform(P.tiles_n), lt = tile - wave_uniform(P.tile0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    TileCtx c;
    c.M = wave_uniform(P.M); c.N = wave_uniform(P.N); c.tm = lt / tiles_n; c.tn = lt - c.tm * tiles_n;
    // staging maps: K-contiguous operand: row = tid / 8 (+ 32), 4 columns from (tid % 8) * 4; NN's B: k = tid / 16 (+ 16), 4 columns from (tid % 16) * 4
    c.sr = tid >> 3; c.sc = (tid & 7) << 2;
    c.br = NN ? tid >> 4 : c.sr; c.bc = NN ? (tid & 15) << 2 : c.sc;
    c.wm = wave >> 1; c.wn = wave & 1; c.li = lane & 31; c.lh = lane >> 5;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int nterm = wave_uniform(P.nterm);
    int parity = 0;
    for (int t = 0; t < nterm; ++t) {
        const Term T = uniform_term(terms[t]);
        const bool fast = T.K % kSlab == 0 && !T.b_bf16 && vec_ok(T.a, T.lda, T.a_bf16) && vec_ok(T.b, T.ldb, 0) && (!NN || (c.tn + 1) * kTile <= c.N);
        if (fast && T.a_bf16) parity = term_fast<NN, true>(T, c, As, Bs, parity, acc);
        else if (fast) parity = term_fast<NN, false>(T, c, As, Bs, parity, acc);
        else parity = term_any<NN>(T, c, As, Bs, parity, acc);
    }
    epilogue_tile(P, c.tm * kTile + c.wm * 32, 1, c.tn * kTile + c.wn * 32 + c.li, c.lh, acc);
}

__device__ __forceinline__ bool pair_ok(const void *base, int64_t ld, int bf16)
{
    return (reinterpret_cast<uintptr_t>(base) & (bf16 ? 3u : 7u)) == 0 && (ld & 1) == 0;
}

// A lane's two neighbouring columns of one operand row, as loaded (`tn_issue`: loads only -- nothing consumes a loaded value, so
// nothing waits for memory here) and as values (`tn_value`, when the step is consumed one step of matrix instructions later).
// PAIR: both columns exist and are aligned -- one 8-byte (fp32) / 4-byte (bf16) load; otherwise two guarded element loads.
template <bool BF16, bool PAIR>
__device__ __forceinline__ void tn_issue(const void *base, int64_t off, bool ok0, bool ok1, unsigned (&r)[2])
{
    if (PAIR) {
        const int64_t o = ok0 ? off : 0;
        if (BF16) {
            r[0] = *(const MDETR_GLOBAL unsigned *)(as_global<unsigned short>(base) + o);
        } else {
            const u32x2 v = *(const MDETR_GLOBAL u32x2 *)(as_global<unsigned>(base) + o);
            r[0] = v.x; r[1] = v.y;
        }
    } else if (BF16) {
        const MDETR_GLOBAL unsigned short *p = as_global<unsigned short>(base);
        r[0] = p[ok0 ? off : 0]; r[1] = p[ok1 ? off + 1 : 0];
    } else {
        const MDETR_GLOBAL unsigned *p = as_global<unsigned>(base);
        r[0] = p[ok0 ? off : 0]; r[1] = p[ok1 ? off + 1 : 0];
    }
}
template <bool BF16, bool PAIR>
__device__ __forceinline__ f32x2 tn_value(const unsigned (&r)[2], bool ok0, bool ok1)
{
    f32x2 v;
    if (!BF16) { v.x = __uint_as_float(r[0]); v.y = __uint_as_float(r[1]); }
    else if (PAIR) { v.x = __uint_as_float(r[0] << 16); v.y = __uint_as_float(r[0] & 0xffff0000u); }
    else { v.x = __uint_as_float(r[0] << 16); v.y = __uint_as_float(r[1] << 16); }
    if (!ok0) v.x = 0.f;
    if (!ok1) v.y = 0.f;
    return v;
}

// this wave's share of a TN contraction: rows [t_begin, t_end), two per matrix instruction (one per lane half); the lane holds the
// columns m, m + 1 of A and n, n + 1 of B of every row: four products (even / odd columns of either operand) per pair of loads.
// Two register sets in turn: while one step's 4 E instructions run, the next step's 2 E loads are in flight.
template <bool ABF, bool BBF, bool PAIR>
__device__ __forceinline__ void tn_contract(const Term &T, int m, int n, int M, int N, int lh, int t_begin, int t_end, f32x16 (&acc)[2][2], float (&cs)[2])
{
    constexpr int E = 4;                                           // a step = 2 E rows: 2 E loads, 4 E matrix instructions
    unsigned ra[3][E][2], rb[3][E][2];                             // three steps in flight (a dependent load costs ~1 us, a step's products 0.46)
    const bool m0 = PAIR || m < M, m1 = PAIR || m + 1 < M, n0 = PAIR || n < N, n1 = PAIR || n + 1 < N;
    auto issue = [&](int t0, unsigned (&xa)[E][2], unsigned (&xb)[E][2]) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int t = t0 + 2 * e + lh;
            const bool ok = t < t_end;
            tn_issue<ABF, PAIR>(T.a, static_cast<int64_t>(t) * T.lda + m, ok && m0, ok && m1, xa[e]);
            tn_issue<BBF, PAIR>(T.b, static_cast<int64_t>(t) * T.ldb + n, ok && n0, ok && n1, xb[e]);
        }
    };
    auto consume = [&](int t0, const unsigned (&xa)[E][2], const unsigned (&xb)[E][2]) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool ok = t0 + 2 * e + lh < t_end;
            const f32x2 av = tn_value<ABF, PAIR>(xa[e], ok && m0, ok && m1), bv = tn_value<BBF, PAIR>(xb[e], ok && n0, ok && n1);
            acc[0][0] = mfma_f32(av.x, bv.x, acc[0][0]);
            acc[0][1] = mfma_f32(av.x, bv.y, acc[0][1]);
            acc[1][0] = mfma_f32(av.y, bv.x, acc[1][0]);
            acc[1][1] = mfma_f32(av.y, bv.y, acc[1][1]);
            cs[0] += av.x;
            cs[1] += av.y;
        }
    };
    // branch-free body: a step beyond the wave's rows loads element 0 and contributes zeros (at most two such steps per wave)
    const int nsteps = (t_end - t_begin + 2 * E - 1) / (2 * E);
    if (nsteps <= 0) return;
    issue(t_begin, ra[0], rb[0]);
    issue(t_begin + 2 * E, ra[1], rb[1]);
    for (int st = 0; st < nsteps; st += 3) {
        const int t0 = t_begin + st * 2 * E;
        issue(t0 + 4 * E, ra[2], rb[2]);
        consume(t0, ra[0], rb[0]);
        issue(t0 + 6 * E, ra[0], rb[0]);
        consume(t0 + 2 * E, ra[1], rb[1]);
        issue(t0 + 8 * E, ra[1], rb[1]);
        consume(t0 + 4 * E, ra[2], rb[2]);
    }
}

// ---- TN ------------------------------------------------------------------------------------------------------------------------
// grid = (tiles, split): workgroup (tile, s) contracts rows [s, s + 1) * ceil(K / split) of its 64 x 64 tile, a quarter per wave,
// summed through LDS in a fixed tree.  split == 1: the tile goes straight to C.  Otherwise it goes to partial s of the workspace
// and sgemm_tn_reduce sums the partials in order: no atomics, bit-identical from run to run.
__global__ __launch_bounds__(64 * kTnWaves)
void sgemm_tn_kernel(const Args a)
{
    __shared__ float red[(kTnWaves / 2) * 64 * 64];               // [slot][64 accumulator registers][64 lanes]
    __shared__ float cred[kTnWaves * 64];                          // column sums per wave
    __shared__ __attribute__((aligned(16))) Args largs;
    const int tile = blockIdx.x;
    const Prob &P = stage_args(a, &largs, tile);
    const Term T = uniform_term(largs.t[wave_uniform(P.term0)]);
    const int PM = wave_uniform(P.M), PN = wave_uniform(P.N), split = wave_uniform(largs.split);
    const int lt = tile - wave_uniform(P.tile0), tm = lt / P.tiles_n, tn = lt - tm * P.tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int m = tm * kTnTile + 2 * li, n = tn * kTnTile + 2 * li;      // the lane's first column of either operand
    // this wave's rows of the contraction: an even count per wave so that the two lane halves pair up inside it
    const int parts = split * kTnWaves;
    const int chunk = ((T.K + 2 * parts - 1) / (2 * parts)) * 2;
    const int t_begin = min(T.K, (static_cast<int>(blockIdx.y) * kTnWaves + wave) * chunk), t_end = min(T.K, t_begin + chunk);

    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = acc[0][1][r] = acc[1][0][r] = acc[1][1][r] = 0.f;
    float cs[2] = {0.f, 0.f};

    // (uniform) both operands' column pairs exist for every lane of the tile and are aligned: one load per pair
    const bool pair = (tm + 1) * kTnTile <= PM && (tn + 1) * kTnTile <= PN && pair_ok(T.a, T.lda, T.a_bf16) && pair_ok(T.b, T.ldb, T.b_bf16);
    if (pair) {
        if (T.a_bf16) {
            if (T.b_bf16) tn_contract<true, true, true>(T, m, n, PM, PN, lh, t_begin, t_end, acc, cs);
            else tn_contract<true, false, true>(T, m, n, PM, PN, lh, t_begin, t_end, acc, cs);
        } else {
            if (T.b_bf16) tn_contract<false, true, true>(T, m, n, PM, PN, lh, t_begin, t_end, acc, cs);
            else tn_contract<false, false, true>(T, m, n, PM, PN, lh, t_begin, t_end, acc, cs);
        }
    } else {
        if (T.a_bf16) {
            if (T.b_bf16) tn_contract<true, true, false>(T, m, n, PM, PN, lh, t_begin, t_end, acc, cs);
            else tn_contract<true, false, false>(T, m, n, PM, PN, lh, t_begin, t_end, acc, cs);
        } else {
            if (T.b_bf16) tn_contract<false, true, false>(T, m, n, PM, PN, lh, t_begin, t_end, acc, cs);
            else tn_contract<false, false, false>(T, m, n, PM, PN, lh, t_begin, t_end, acc, cs);
        }
    }
    // the waves' partial tiles, summed pairwise in a fixed tree: wave w + half hands its tile to wave w (deterministic)
    cs[0] += __shfl_xor(cs[0], 32);
    cs[1] += __shfl_xor(cs[1], 32);
    if (lh == 0) { cred[wave * 64 + 2 * li] = cs[0]; cred[wave * 64 + 2 * li + 1] = cs[1]; }
#pragma unroll 1
    for (int half = kTnWaves / 2; half >= 1; half >>= 1) {
        __syncthreads();
        if (wave >= half && wave < 2 * half) {
            float *slot = red + (wave - half) * 64 * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int r = 0; r < 16; ++r) slot[(q * 16 + r) * 64 + lane] = acc[q >> 1][q & 1][r];
            }
        }
        __syncthreads();
        if (wave < half) {
            const float *slot = red + wave * 64 * 64;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q >> 1][q & 1][r] += slot[(q * 16 + r) * 64 + lane];
            }
        }
    }
    if (wave != 0) return;
    float sum = 0.f;                                               // lane l sums column l of the tile over the waves, in wave order
#pragma unroll
    for (int w = 0; w < kTnWaves; ++w) sum += cred[w * 64 + lane];
    const int mm = tm * kTnTile + lane;
    if (split == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            epilogue_tile(P, tm * kTnTile + (q >> 1), 2, tn * kTnTile + 2 * li + (q & 1), lh, acc[q >> 1][q & 1]);
        if (P.colsum && tn == 0 && mm < PM) as_global_rw<float>(P.colsum)[mm] = sum;
        return;
    }
    MDETR_GLOBAL float *part = as_global_rw<float>(largs.ws) + static_cast<int64_t>(blockIdx.y) * largs.flat + P.foff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = tn * kTnTile + 2 * li + (q & 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * kTnTile + (q >> 1) + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lh);
            if (row < PM && col < PN) part[static_cast<int64_t>(row) * PN + col] = acc[q >> 1][q & 1][r];
        }
    }
    if (tn == 0 && mm < PM) part[static_cast<int64_t>(PM) * PN + mm] = sum;
}

// C = the sum of the `split` partials, in order; one thread per element of the flat result (M x N values + M column sums per problem)
__global__ __launch_bounds__(256)
void sgemm_tn_reduce(const Args a)
{
    __shared__ __attribute__((aligned(16))) Args largs;
    {
        const unsigned *src = reinterpret_cast<const unsigned *>(&a);
        unsigned *dst = reinterpret_cast<unsigned *>(&largs);
        for (unsigned i = threadIdx.x; i < sizeof(Args) / 4; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const int f = blockIdx.x * 256 + threadIdx.x, flat = largs.flat;
    if (f >= flat) return;
    int pi = 0;
    for (int i = 1; i < largs.nprob; ++i)
        if (f >= largs.p[i].foff) pi = i;
    const Prob &P = largs.p[pi];
    const MDETR_GLOBAL float *part = as_global<float>(largs.ws) + f;
    float v = 0.f;
    for (int s2 = 0; s2 < largs.split; ++s2) v += part[static_cast<int64_t>(s2) * flat];
    const int e = f - P.foff, mn = P.M * P.N;
    if (e < mn) {
        const int row = e / P.N, col = e - row * P.N;
        const int64_t o = static_cast<int64_t>(row) * P.ldc + col;
        if (P.flags & kFlagCBf16) as_global_rw<__bf16>(P.c)[o] = static_cast<__bf16>(v);
        else as_global_rw<float>(P.c)[o] = v;
    } else if (P.colsum) {
        as_global_rw<float>(P.colsum)[e - mn] = v;
    }
}

// how many parts the contraction of a TN group is cut into: enough workgroups to fill the chip, parts of >= 64 rows per wave
int tn_split(const mdetr_sgemm_problem *p, int nprob)
{
    int tiles = 0, kmin = 1 << 30;
    for (int i = 0; i < nprob; ++i) {
        tiles += ((p[i].m + kTnTile - 1) / kTnTile) * ((p[i].n + kTnTile - 1) / kTnTile);
        kmin = p[i].term[0].k < kmin ? p[i].term[0].k : kmin;
    }
    int s = (768 + tiles - 1) / tiles;
    const int by_rows = kmin / (64 * kTnWaves);
    s = s < by_rows ? s : by_rows;
    return s < 1 ? 1 : (s > kTnMaxSplit ? kTnMaxSplit : s);
}

int64_t tn_flat(const mdetr_sgemm_problem *p, int nprob)
{
    int64_t f = 0;
    for (int i = 0; i < nprob; ++i) f += static_cast<int64_t>(p[i].m) * p[i].n + p[i].m;
    return f;
}

bool dtype_ok(int d) { return d == MDETR_F32 || d == MDETR_BF16; }

}  // namespace

const char *sgemm_check(int mode, const mdetr_sgemm_problem *p, int nprob)
{
    if (mode != MDETR_SGEMM_NT && mode != MDETR_SGEMM_NN && mode != MDETR_SGEMM_TN) return "unknown mode";
    if (!p || nprob <= 0 || nprob > MDETR_SGEMM_MAX_PROBLEMS) return "1 .. MDETR_SGEMM_MAX_PROBLEMS problems per group";
    int terms = 0;
    for (int i = 0; i < nprob; ++i) {
        const mdetr_sgemm_problem &q = p[i];
        if (q.m <= 0 || q.n <= 0 || q.nterm <= 0 || q.nterm > MDETR_SGEMM_MAX_TERMS) return "bad shape (m, n > 0; 1 .. MDETR_SGEMM_MAX_TERMS terms)";
        if (!q.c || q.ldc < q.n || !dtype_ok(q.c_dtype)) return "bad result (null, row stride < n, or element type)";
        if (q.relu_cols < 0 || q.relu_cols > q.n) return "relu_cols outside [0, n]";
        if (q.mask && q.ldm < q.n) return "mask row stride < n";
        if (q.res && (q.ldr < q.n || !dtype_ok(q.res_dtype))) return "bad res (row stride < n, or element type)";
        if (mode == MDETR_SGEMM_TN && (q.nterm != 1 || q.bias || q.mask || q.res || q.relu_cols)) return "TN: one term, no bias / relu / mask / res";
        if (mode != MDETR_SGEMM_TN && q.colsum) return "colsum belongs to TN";
        for (int t = 0; t < q.nterm; ++t) {
            const mdetr_sgemm_term &x = q.term[t];
            if (!x.a || !x.b || x.k <= 0 || !dtype_ok(x.a_dtype) || !dtype_ok(x.b_dtype)) return "bad term (null operand, k <= 0, or element type)";
            const int64_t a_cols = mode == MDETR_SGEMM_TN ? q.m : x.k, b_cols = mode == MDETR_SGEMM_NT ? x.k : q.n;
            if (x.lda < a_cols || x.ldb < b_cols) return "operand row stride shorter than its row";
        }
        terms += q.nterm;
    }
    if (terms > kMaxTerms) return "too many terms in one group";
    if (mode == MDETR_SGEMM_TN && tn_flat(p, nprob) >= (1ll << 31)) return "TN group too large";
    return nullptr;
}

int64_t sgemm_workspace_bytes(int mode, const mdetr_sgemm_problem *p, int nprob)
{
    if (mode != MDETR_SGEMM_TN) return 0;
    const int split = tn_split(p, nprob);
    return split > 1 ? static_cast<int64_t>(split) * tn_flat(p, nprob) * 4 : 0;
}

hipError_t sgemm_launch(int mode, const mdetr_sgemm_problem *p, int nprob, void *workspace, hipStream_t st)
{
    Args a;
    a.split = 1; a.flat = 0; a.ws = nullptr;
    int tiles = 0, terms = 0;
    int64_t foff = 0;
    double flop = 0.0, bytes = 0.0;
    const int tsz = mode == MDETR_SGEMM_TN ? kTnTile : kTile;
    for (int i = 0; i < nprob; ++i) {
        const mdetr_sgemm_problem &q = p[i];
        Prob &P = a.p[i];
        P.c = q.c; P.bias = q.bias; P.colsum = q.colsum; P.mask = q.mask; P.res = q.res;
        P.ldc = q.ldc; P.ldm = q.ldm; P.ldr = q.ldr;
        P.M = q.m; P.N = q.n; P.relu_cols = q.relu_cols; P.term0 = terms; P.nterm = q.nterm;
        P.tile0 = tiles; P.tiles_n = (q.n + tsz - 1) / tsz;
        P.flags = (q.c_dtype == MDETR_BF16 ? kFlagCBf16 : 0) | (q.res && q.res_dtype == MDETR_BF16 ? kFlagResBf16 : 0);
        P.foff = static_cast<int>(foff); P.pad_ = 0;
        foff += static_cast<int64_t>(q.m) * q.n + q.m;
        tiles += ((q.m + tsz - 1) / tsz) * P.tiles_n;
        bytes += static_cast<double>(q.m) * q.n * (q.c_dtype == MDETR_BF16 ? 2.0 : 4.0) * (1.0 + (q.res ? 1.0 : 0.0)) + (q.mask ? 4.0 * q.m * q.n : 0.0);
        for (int t = 0; t < q.nterm; ++t) {
            const mdetr_sgemm_term &x = q.term[t];
            Term &T = a.t[terms++];
            T.a = x.a; T.b = x.b; T.lda = x.lda; T.ldb = x.ldb; T.K = x.k;
            T.a_bf16 = x.a_dtype == MDETR_BF16; T.b_bf16 = x.b_dtype == MDETR_BF16; T.pad_ = 0;
            flop += 2.0 * q.m * q.n * x.k;
            bytes += static_cast<double>(x.k) * (q.m * (T.a_bf16 ? 2.0 : 4.0) + q.n * (T.b_bf16 ? 2.0 : 4.0));
        }
    }
    a.nprob = nprob;
    ProfileScope prof(21, tiles, st, flop / 1e6, bytes / 1e3);
    if (mode == MDETR_SGEMM_TN) {
        a.split = tn_split(p, nprob);
        a.flat = static_cast<int>(tn_flat(p, nprob));
        a.ws = static_cast<float *>(workspace);
        hipLaunchKernelGGL(sgemm_tn_kernel, dim3(tiles, a.split), dim3(64 * kTnWaves), 0, st, a);
        if (a.split > 1) hipLaunchKernelGGL(sgemm_tn_reduce, dim3((a.flat + 255) / 256), dim3(256), 0, st, a);
    }
    else if (mode == MDETR_SGEMM_NN) hipLaunchKernelGGL(sgemm_kernel<true>, dim3(tiles), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(sgemm_kernel<false>, dim3(tiles), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace mdetr
