// monodetr_amd/csrc/token_gemm.hip -- y[T, N] = x[T, K] W^T + b  (+ ReLU) for tall token matrices, bf16,
// with the WEIGHT RESIDENT IN LDS.
//
// Why: the transformer's linear layers act on T = 81 600 tokens with K = 256 (21 forward + 21
// input-gradient products per training step).  Such a product is 10.7 GFLOP against 84 MB of
// compulsory traffic -- memory-bound, floor ~10 us at 8 TB/s -- and the library kernel hipBLASLt picks
// (MT64x64x128) takes 41 us = 2 TB/s (profiles/r01h_bench_bf16_steady_kernel_stats.csv).  A 256 x 256
// bf16 weight is 128 KiB: it fits the 160 KiB LDS of a CU whole.
//
// Layout of the computation (the fragment conventions are those of attn.hip, validated there):
//   * a workgroup = 4 waves loads W[n0 : n0 + NB*32, 0:K] once into LDS, row-major, rows padded by 8 bf16
//     (264 elements = 132 dwords = 4 mod 64: the 16 rows of a ds_read_b128 lane group hit 16 distinct
//     4-bank slots);
//   * each wave then walks 32-token tiles (grid-stride): the tile's 32 x K inputs are fetched with
//     coalesced 16-byte loads into registers, four K-slabs of 64, and passed through a wave-private LDS
//     slab [32][72] so that a lane can read ITS token's 8 contiguous k values as the B operand;
//     the next tile's slab s is requested right after slab s of the current tile has been written to
//     LDS, so a whole tile of loads (16 KiB per wave) is in flight during the matrix work;
//   * products are issued transposed, Y^T[n][token] = W[n][:] . X[token][:]  (A = weight rows from
//     LDS, B = the lane's token), with v_mfma_f32_32x32x16_bf16: a lane's accumulator then holds four
//     consecutive output features of ITS token per register quad -> 8-byte stores, bias / ReLU applied
//     in registers.
// Algorithmic bytes = 2 (T K + T N) + 2 N K; MFMA work is 4-5x below the memory time at these shapes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <mdetr_wave.h>

#include "token_gemm.h"

namespace mdetr {
namespace {

constexpr int kWavesG = 4;
constexpr int kSlabK = 64;                 // k values per slab
constexpr int kSlabPad = kSlabK + 8;       // 72 bf16 = 36 dwords per slab row (conflict-free b128 reads, see above)

// accumulator register r of a lane holds row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31 of the
// 32 x 32 result (mdetr_wave.h, attn.hip acc_row)

// K = contraction length (multiple of 64), NB = number of 32-wide output blocks held by a workgroup
template <int K, int NB, bool RELU>
__global__ __launch_bounds__(kWavesG * 64)
void token_gemm_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w, const __bf16 *__restrict__ bias,
                       __bf16 *__restrict__ y, int64_t T, int N, int64_t ldx, int64_t ldy)
{
    constexpr int KP = K + 8;                                    // padded weight row
    constexpr int NS = K / kSlabK;                               // slabs per tile
    MDETR_DYNAMIC_LDS(unsigned char, smem_raw);
    __bf16 *Ws = reinterpret_cast<__bf16 *>(smem_raw);           // [NB*32][KP]
    __bf16 *slabs = Ws + NB * 32 * KP;                           // [4 waves][32][kSlabPad]
    float *bias_s = reinterpret_cast<float *>(slabs + kWavesG * 32 * kSlabPad);   // [NB*32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    const int n0 = blockIdx.y * NB * 32;

    // ---- weight (and bias) into LDS, once: 16-byte chunks, consecutive threads -> consecutive chunks of a row
    for (int c = threadIdx.x; c < NB * 32 * (K / 8); c += kWavesG * 64) {
        const int row = c / (K / 8), col = (c - row * (K / 8)) * 8;
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = static_cast<__bf16>(0.f);
        if (n0 + row < N) v = *reinterpret_cast<const bf16x8 *>(w + static_cast<int64_t>(n0 + row) * K + col);
        *reinterpret_cast<bf16x8 *>(Ws + row * KP + col) = v;
    }
    for (int c = threadIdx.x; c < NB * 32; c += kWavesG * 64)
        bias_s[c] = (bias && n0 + c < N) ? static_cast<float>(bias[n0 + c]) : 0.f;
    __syncthreads();

    __bf16 *slab = slabs + wave * 32 * kSlabPad;
    const int64_t tiles = (T + 31) / 32;
    const int64_t wave_id = static_cast<int64_t>(blockIdx.x) * kWavesG + wave, wave_n = static_cast<int64_t>(gridDim.x) * kWavesG;

    // staging map of one slab (32 rows x 64 k = 256 chunks of 16 B): chunk c = lane + 64 j -> row c / 8, piece c % 8
    bf16x8 stage[NS][4];
    auto request = [&](int64_t tile, int s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j, row = c >> 3, piece = c & 7;
            const int64_t t = tile * 32 + row;
            bf16x8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = static_cast<__bf16>(0.f);
            if (t < T) v = *reinterpret_cast<const bf16x8 *>(x + t * ldx + s * kSlabK + piece * 8);
            stage[s][j] = v;
        }
    };

    int64_t tile = wave_id;
    if (tile < tiles) {
#pragma unroll
        for (int s = 0; s < NS; ++s) request(tile, s);
    }
    for (; tile < tiles; tile += wave_n) {
        f32x16 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
        const int64_t next = tile + wave_n;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            wave_sync();                                         // the previous slab's fragment reads are done
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = lane + 64 * j, row = c >> 3, piece = c & 7;
                *reinterpret_cast<bf16x8 *>(slab + row * kSlabPad + piece * 8) = stage[s][j];
            }
            if (next < tiles) request(next, s);                  // next tile's slab s: in flight during the products below
            wave_sync();
#pragma unroll
            for (int ks = 0; ks < kSlabK / 16; ++ks) {
                // B operand: this lane's token (row lane & 31 of the slab), 8 contiguous k
                const bf16x8 xb = *reinterpret_cast<const bf16x8 *>(slab + (lane & 31) * kSlabPad + ks * 16 + half * 8);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    // A operand: weight row n = 32 b + (lane & 31), the same 8 k values of the global K axis
                    const bf16x8 wa = *reinterpret_cast<const bf16x8 *>(Ws + (b * 32 + (lane & 31)) * KP + s * kSlabK + ks * 16 + half * 8);
                    acc[b] = mfma_bf16(wa, xb, acc[b]);          // Y^T[n][token]
                }
            }
        }
        // ---- epilogue: lane = token (lane & 31); register quad g holds features 32 b + 8 g + 4 half + 0..3
        const int64_t t = tile * 32 + (lane & 31);
        if (t < T) {
            __bf16 *yr = y + t * ldy + n0 + 4 * half;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nn = b * 32 + 8 * g + 4 * half;     // first of the four features, relative to n0
                    if (n0 + nn < N) {                            // N is a multiple of 4: a quad is in or out as a whole
                        bf16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float v = acc[b][4 * g + i] + bias_s[nn + i];
                            if (RELU) v = v > 0.f ? v : 0.f;
                            o[i] = static_cast<__bf16>(v);
                        }
                        *reinterpret_cast<bf16x4 *>(yr + b * 32 + 8 * g) = o;
                    }
                }
            }
        }
    }
}

// ---- the "direct" form (MDETR_TOKEN_GEMM_DIRECT=1) ------------------------------------------------------------------------------
// What the listing of the kernel above shows (tools/isa_blocks.py): (1) its weight prologue is ONE 16-byte load per loop trip,
// waited for and stored -- 32 dependent round trips to L2 before the first product; (2) with 256 VGPRs taken (8 accumulators +
// a tile of staged inputs) the compiler reuses one register quad for every weight fragment: ds_read -> s_waitcnt 0 -> MFMA, 128
// times per tile at one wave per SIMD, i.e. an LDS latency per product; (3) the inputs take a round trip through LDS only to
// change which lane holds which 16 bytes.  Here instead:
//   * the B operand of v_mfma_f32_32x32x16_bf16 for lane (token = lane & 31, half = lane >> 5) at k-step ks is the 8
//     CONTIGUOUS values x[token][16 ks + 8 half ...]: the lane loads them itself, 16 bytes, straight from global memory.  The
//     four loads of a 64-value slab are issued back to back (they touch the same 32 cache lines), a slab's registers are
//     refilled for the wave's NEXT tile as soon as its four k-steps are consumed: a whole tile (16 KiB per wave) stays in flight.
//   * the weight fragments of k-step ks + 1 are read from LDS while the products of k-step ks issue (two register sets);
//   * NB = 4 output blocks per workgroup at K = 256: 68 KB of LDS and <= 256 registers, so TWO workgroups = two waves per SIMD
//     share a CU (the column halves of N = 256 walk the same token tiles at the same time: the second reader of x hits L2);
//   * the weight prologue keeps 16 loads in flight per lane (and the first tile's operands are requested ahead of it).
template <int K, int NB, bool RELU>
__global__ __launch_bounds__(kWavesG * 64, (NB * 32 * (K + 8) * 2 + NB * 32 * 4 <= 80 * 1024) ? 2 : 1)     // two workgroups per CU where the LDS allows
void token_gemm_direct_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w, const __bf16 *__restrict__ bias,
                              __bf16 *__restrict__ y, int64_t T, int N, int64_t ldx, int64_t ldy, int gx, int ny)
{
    constexpr int KP = K + 8;                                    // padded weight row (see above)
    constexpr int NS = K / kSlabK, KS = K / 16;                  // slabs, k-steps per tile
    constexpr int kChunks = NB * 32 * (K / 8), kPer = kChunks / (kWavesG * 64);
    static_assert(kChunks % (kWavesG * 64) == 0, "weight chunks divide over the workgroup");
    MDETR_DYNAMIC_LDS(unsigned char, smem_raw);
    __bf16 *Ws = reinterpret_cast<__bf16 *>(smem_raw);           // [NB*32][KP]
    float *bias_s = reinterpret_cast<float *>(Ws + NB * 32 * KP);   // [NB*32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    // 1-D grid of gx * ny workgroups, gx a multiple of 8.  Workgroup ids go round-robin over the 8 XCDs (id % 8), each with its own
    // L2: the ny column blocks of one token range are ids 8 apart -- the same XCD, neighbouring dispatch slots -- so that the
    // second reader of an x tile finds it in that L2 instead of fetching it from memory again.
    const int id = blockIdx.x, grp = id >> 3;
    const int col = grp % ny, bx = (grp / ny) * 8 + (id & 7);    // column block, token-range index (< gx)
    const int n0 = col * NB * 32;
    const int64_t tiles = (T + 31) / 32;
    if (static_cast<int64_t>(bx) * kWavesG >= tiles) return;     // (gx was rounded up to a multiple of 8)

    const int64_t wave_id = static_cast<int64_t>(bx) * kWavesG + wave, wave_n = static_cast<int64_t>(gx) * kWavesG;
    // the first tile's operands are requested BEFORE the weight is staged: their latency hides behind the prologue
    bf16x8 xb[KS];                                               // the lane's B operands of one tile
    auto request = [&](int64_t tile_, int s) __attribute__((always_inline)) {
        // (rows beyond T read row T - 1: their products are computed and never stored -- no branch, no zero fill)
        const int64_t t = tile_ * 32 + (lane & 31), tc = t < T ? t : T - 1;
        const __bf16 *p = x + tc * ldx + s * kSlabK + half * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) xb[s * 4 + q] = *reinterpret_cast<const bf16x8 *>(p + q * 16);
    };

    int64_t tile = wave_id;
    if (tile < tiles) {
#pragma unroll
        for (int s = 0; s < NS; ++s) request(tile, s);
    }

    // ---- weight (and bias) into LDS, once: batches of kBatch independent 16-byte loads per lane
    constexpr int kBatch = 16;
#pragma unroll
    for (int c0 = 0; c0 < kPer; c0 += kBatch) {
        bf16x8 v[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            if (c0 + u < kPer) {                                  // (compile time)
                const int c = (c0 + u) * (kWavesG * 64) + static_cast<int>(threadIdx.x);
                const int row = c / (K / 8), kc = (c - row * (K / 8)) * 8;
                // rows beyond N read row N - 1 and are zeroed at the store: unconditional loads, nothing between them
                const int rc = n0 + row < N ? n0 + row : N - 1;
                v[u] = *reinterpret_cast<const bf16x8 *>(w + static_cast<int64_t>(rc) * K + kc);
            }
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            if (c0 + u < kPer) {
                const int c = (c0 + u) * (kWavesG * 64) + static_cast<int>(threadIdx.x);
                const int row = c / (K / 8), kc = (c - row * (K / 8)) * 8;
                if (n0 + row >= N) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[u][i] = static_cast<__bf16>(0.f);
                }
                *reinterpret_cast<bf16x8 *>(Ws + row * KP + kc) = v[u];
            }
        }
    }
    for (int c = threadIdx.x; c < NB * 32; c += kWavesG * 64)
        bias_s[c] = (bias && n0 + c < N) ? static_cast<float>(bias[n0 + c]) : 0.f;
    __syncthreads();

    const __bf16 *wl = Ws + (lane & 31) * KP + half * 8;         // this lane's fragment of block b, k-step ks: + b * 32 * KP + 16 ks

    for (; tile < tiles; tile += wave_n) {
        f32x16 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
        const int64_t next = tile + wave_n;
        bf16x8 wa[2][NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) wa[0][b] = *reinterpret_cast<const bf16x8 *>(wl + b * 32 * KP);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int b = 0; b < NB; ++b) wa[(ks + 1) & 1][b] = *reinterpret_cast<const bf16x8 *>(wl + b * 32 * KP + (ks + 1) * 16);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[b] = mfma_bf16(wa[ks & 1][b], xb[ks], acc[b]);       // Y^T[n][token]
            if ((ks & 3) == 3 && next < tiles) request(next, ks >> 2);                           // this slab's registers are free again
            // (a fence for the COMPILER only: left alone it hoists all NB x KS fragment reads -- they do not depend on the tile --
            // to the top of the tile and spills them: 644 bytes of scratch per lane)
            __asm__ volatile("" ::: "memory");
            // ... and the order within the step: one fragment read of step ks + 1 ahead of each product of step ks
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // one LDS read
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // one MFMA
            }
        }
        // ---- epilogue: lane = token (lane & 31); register quad g holds features 32 b + 8 g + 4 half + 0..3
        const int64_t t = tile * 32 + (lane & 31);
        if (t < T) {
            __bf16 *yr = y + t * ldy + n0 + 4 * half;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nn = b * 32 + 8 * g + 4 * half;
                    if (n0 + nn < N) {
                        bf16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float v = acc[b][4 * g + i] + bias_s[nn + i];
                            if (RELU) v = v > 0.f ? v : 0.f;
                            o[i] = static_cast<__bf16>(v);
                        }
                        *reinterpret_cast<bf16x4 *>(yr + b * 32 + 8 * g) = o;
                    }
                }
            }
        }
    }
}

// ---- the "weight in registers" form (MDETR_TOKEN_GEMM_DIRECT=2) -------------------------------------------------------------
// The two forms above hold the weight in LDS and give every wave its own tokens: each product reads a 1 KB weight fragment from
// LDS, the 135 KB weight block allows one workgroup of four waves per CU (one wave per SIMD: nothing hides a wait), and the
// direct form fetches x in 16-byte pieces of a row per instruction.  Here the roles are swapped:
//   * a workgroup = 8 waves owns up to 256 output features, wave v the 32 features n0 + 32 v ...: its weight slice [32][K] is 16 KB
//     at K = 256 -- K / 16 register quads per lane (the A operands of v_mfma_f32_32x32x16_bf16), loaded ONCE;
//   * the workgroup walks tiles of TT = 64 (or 32) tokens: the tile's [TT][K] inputs come in with full-row coalesced 16-byte
//     loads (a wave covers two whole 512-byte rows per instruction), go through registers into one of two LDS buffers
//     (rows padded by 8 bf16: conflict-free ds_read_b128), and ALL eight waves read their B operands from there -- x is
//     fetched from memory once per 256 output features, 32 KB per tile and workgroup;
//   * two register stages: the loads of tile i + 2 are issued before the products of tile i, the stage holding tile i + 1 is
//     written to LDS after them -- one barrier per tile, 1.5 - 2 tiles (48 - 64 KB per CU) in flight;
//   * 1 275 tiles of 64 tokens at T = 81 600 = 4.98 per CU: five rounds with no tail (256-token tiles: 319 over 256 CUs = two
//     rounds for 1.25 rounds of work).
// LDS 68 KB, ~170 registers at two waves per SIMD.  Same products in the same order per output element as the other forms.
constexpr int kWavesW = 8;

template <int K, int TT, bool RELU, bool YS>
__global__ __launch_bounds__(kWavesW * 64, 2)
void token_gemm_ws_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w, const __bf16 *__restrict__ bias,
                          __bf16 *__restrict__ y, int64_t T, int N, int64_t ldx, int64_t ldy, int gx, int ny, int ablate, int wide)
{
    // ablate (MDETR_TOKEN_GEMM_ABLATE, developer timing runs only -- the results are then wrong): bit 0 = no output stores,
    // bit 1 = every tile re-reads the workgroup's first tile (inputs from L2 instead of memory), bit 2 = no products
    constexpr int KP = K + 8, KS = K / 16, TB = TT / 32;
    constexpr int kChunks = TT * (K / 8), kPer = (kChunks + kWavesW * 64 - 1) / (kWavesW * 64);      // 16-byte pieces of a tile, per thread
    MDETR_DYNAMIC_LDS(unsigned char, smem_raw);
    constexpr int YP = kWavesW * 32 + 8;                          // padded row of the output tile
    __bf16 *Xs = reinterpret_cast<__bf16 *>(smem_raw);           // [2][TT][KP]
    __bf16 *Ysm = Xs + 2 * TT * KP;                              // YS: [2][TT][YP] -- the output tile, written back in whole rows
    float *bias_s = reinterpret_cast<float *>(Ysm + (YS ? 2 * TT * YP : 0));  // [256]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
    // 1-D grid of gx * ny workgroups, gx a multiple of 8: the ny column blocks of one token range share an XCD (see the direct form)
    const int id = blockIdx.x, grp = id >> 3;
    const int col = grp % ny, bx = (grp / ny) * 8 + (id & 7);
    const int n0 = col * kWavesW * 32;
    const int64_t tiles = (T + TT - 1) / TT;
    if (bx >= tiles) return;

    // one tile's inputs: piece c = thread + 512 p -> row c / (K / 8), 16-byte piece c % (K / 8); rows beyond T re-read row T - 1
    // (their products are never stored)
    auto request = [&](bf16x8 (&st_)[kPer], int64_t tile_) __attribute__((always_inline)) {
#pragma unroll
        for (int p_ = 0; p_ < kPer; ++p_) {
            const int c = static_cast<int>(threadIdx.x) + kWavesW * 64 * p_;
            if (kChunks % (kWavesW * 64) == 0 || c < kChunks) {
                const int row = c / (K / 8), piece = c - row * (K / 8);
                const int64_t t = ((ablate & 2) ? static_cast<int64_t>(bx) : tile_) * TT + row, tc = t < T ? t : T - 1;
                st_[p_] = *reinterpret_cast<const bf16x8 *>(x + tc * ldx + piece * 8);
            }
        }
    };
    auto deposit = [&](const bf16x8 (&st_)[kPer], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int p_ = 0; p_ < kPer; ++p_) {
            const int c = static_cast<int>(threadIdx.x) + kWavesW * 64 * p_;
            if (kChunks % (kWavesW * 64) == 0 || c < kChunks) {
                const int row = c / (K / 8), piece = c - row * (K / 8);
                *reinterpret_cast<bf16x8 *>(Xs + (buf * TT + row) * KP + piece * 8) = st_[p_];
            }
        }
    };

    bf16x8 st0[kPer], st1[kPer];
    int64_t tile = bx;
    request(st0, tile);                                          // tile 0 of this workgroup: requested ahead of the weight
    if (tile + gx < tiles) request(st1, tile + gx);

    // ---- this wave's weight slice into registers: fragment ks = W[n0 + 32 wave + (lane & 31)][16 ks + 8 half ...]
    const int nrow = n0 + wave * 32 + (lane & 31);
    const bool live = nrow < N;
    const __bf16 *wr = w + static_cast<int64_t>(live ? nrow : N - 1) * K + half * 8;
    bf16x8 wa[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        wa[ks] = *reinterpret_cast<const bf16x8 *>(wr + ks * 16);
        if (!live) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wa[ks][i] = static_cast<__bf16>(0.f);
        }
    }
    for (int c = threadIdx.x; c < kWavesW * 32; c += kWavesW * 64)
        bias_s[c] = (bias && n0 + c < N) ? static_cast<float>(bias[n0 + c]) : 0.f;

    deposit(st0, 0);
    if (tile + 2 * static_cast<int64_t>(gx) < tiles) request(st0, tile + 2 * static_cast<int64_t>(gx));
    __syncthreads();

    // products and epilogue of the tile in LDS buffer `buf`
    auto compute = [&](int64_t tile_, int buf, int ybuf) __attribute__((always_inline)) {
        f32x16 acc[TB];
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
        const __bf16 *xl = Xs + (buf * TT + (lane & 31)) * KP + half * 8;      // this lane's token of block b: + b * 32 * KP; k-step: + 16 ks
        if (!(ablate & 4)) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int b = 0; b < TB; ++b) {
                    const bf16x8 xb = *reinterpret_cast<const bf16x8 *>(xl + b * 32 * KP + ks * 16);
                    acc[b] = mfma_bf16(wa[ks], xb, acc[b]);      // Y^T[n][token]
                }
            }
        }
        // lane = token (lane & 31) of block b; register quad g holds features 32 wave + 8 g + 4 half + 0..3
        if (YS) {
            // through LDS: the tile leaves in whole 512-byte rows after the barrier (store_tile); 8-byte pieces of 32 rows per store
            // instruction cost 11 of the form's 28 us at [81 600, 256] x [256, 256] (profiles/r04tga_tokenbench_ablate.json)
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                __bf16 *yl = Ysm + (ybuf * TT + b * 32 + (lane & 31)) * YP + wave * 32 + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[b][4 * g + i] + bias_s[wave * 32 + 8 * g + 4 * half + i];
                        if (RELU) v = v > 0.f ? v : 0.f;
                        o[i] = static_cast<__bf16>(v);
                    }
                    *reinterpret_cast<bf16x4 *>(yl + 8 * g) = o;
                }
            }
            return;
        }
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int64_t t = tile_ * TT + b * 32 + (lane & 31);
            if (t < T && !(ablate & 1)) {
                __bf16 *yr = y + t * ldy + n0 + wave * 32 + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nn = wave * 32 + 8 * g + 4 * half;
                    if (n0 + nn < N) {                            // N is a multiple of 4: a quad is in or out as a whole
                        bf16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float v = acc[b][4 * g + i] + bias_s[nn + i];
                            if (RELU) v = v > 0.f ? v : 0.f;
                            o[i] = static_cast<__bf16>(v);
                        }
                        *reinterpret_cast<bf16x4 *>(yr + 8 * g) = o;
                    }
                }
            }
        }
    };

    // the output tile of LDS buffer `ybuf`: piece c = thread + 512 p -> row c / 32, 16 bytes at feature 8 (c % 32)
    auto store_tile = [&](int64_t tile_, int ybuf) __attribute__((always_inline)) {
        if (!YS || (ablate & 1)) return;
#pragma unroll
        for (int p_ = 0; p_ < TT * 32 / (kWavesW * 64); ++p_) {
            const int c = static_cast<int>(threadIdx.x) + kWavesW * 64 * p_;
            const int row = c >> 5, piece = c & 31;
            const int64_t t = tile_ * TT + row;
            const int n = n0 + piece * 8;
            if (t < T && n < N) {                                 // N is a multiple of 8: a piece is in or out as a whole
                const __bf16 *src = Ysm + (ybuf * TT + row) * YP + piece * 8;
                __bf16 *dst = y + t * ldy + n;
                if (wide) {
                    *reinterpret_cast<bf16x8 *>(dst) = *reinterpret_cast<const bf16x8 *>(src);
                } else {                                          // rows that are only 8-byte aligned
                    *reinterpret_cast<bf16x4 *>(dst) = *reinterpret_cast<const bf16x4 *>(src);
                    *reinterpret_cast<bf16x4 *>(dst + 4) = *reinterpret_cast<const bf16x4 *>(src + 4);
                }
            }
        }
    };

    // two tiles per trip: stage st1 holds tile i + 1 and st0 tile i + 2 on entry (whichever exist).  One barrier per tile: it
    // publishes the next tile's inputs AND this tile's outputs; the output buffers alternate, so a wave can only overwrite one
    // after the barrier that follows the row stores out of it.
    const int64_t step = gx;
    for (;; tile += 2 * step) {
        compute(tile, 0, 0);
        const bool more = tile + step < tiles;
        if (more) {
            deposit(st1, 1);                                     // tile i + 1 (buffer 1 was last read before the previous barrier)
            if (tile + 3 * step < tiles) request(st1, tile + 3 * step);
        }
        if (YS || more) __syncthreads();
        store_tile(tile, 0);
        if (!more) break;
        compute(tile + step, 1, 1);
        const bool more2 = tile + 2 * step < tiles;
        if (more2) {
            deposit(st0, 0);                                     // tile i + 2
            if (tile + 4 * step < tiles) request(st0, tile + 4 * step);
        }
        if (YS || more2) __syncthreads();
        store_tile(tile + step, 1);
        if (!more2) break;
    }
}

template <int K, int TT, bool RELU, bool YS>
hipError_t launch_ws(const void *x, const void *w, const void *bias, void *y, int64_t T, int N, int64_t ldx, int64_t ldy,
                     hipStream_t st)
{
    constexpr size_t lds = static_cast<size_t>(2) * TT * (K + 8) * 2 + (YS ? static_cast<size_t>(2) * TT * (kWavesW * 32 + 8) * 2 : 0) +
                           kWavesW * 32 * 4;
    static_assert(lds <= 160 * 1024, "the tiles do not fit the LDS");
    auto kern = token_gemm_ws_kernel<K, TT, RELU, YS>;
    static bool attr_set[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = -1;
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    const int64_t tiles = (T + TT - 1) / TT;
    const int ny = (N + kWavesW * 32 - 1) / (kWavesW * 32);
    int64_t gx = tiles;
    // one workgroup per CU (two waves per SIMD); MDETR_TOKEN_GEMM_WS_PER_CU=2 (A/B runs): two, where the 32-token form's 69 KB of
    // LDS and <= 128 registers allow it -- the products and stores of one workgroup beside the loads of the other
    int per_cu = 1;
    if (const char *pc = getenv("MDETR_TOKEN_GEMM_WS_PER_CU")) per_cu = (atoi(pc) == 2 && lds <= 80 * 1024) ? 2 : 1;
    const int64_t cap = (256 * per_cu) / ny > 0 ? (256 * per_cu) / ny : 1;
    if (gx > cap) gx = cap;
    gx = (gx + 7) / 8 * 8;                                       // whole rounds over the XCDs (idle workgroups leave at once)
    const char *ab = getenv("MDETR_TOKEN_GEMM_ABLATE");
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(gx * ny)), dim3(kWavesW * 64), lds, st,
                       static_cast<const __bf16 *>(x), static_cast<const __bf16 *>(w), static_cast<const __bf16 *>(bias),
                       static_cast<__bf16 *>(y), T, N, ldx, ldy, static_cast<int>(gx), ny, ab ? atoi(ab) : 0,
                       ((reinterpret_cast<uintptr_t>(y) & 15) == 0 && ldy % 8 == 0) ? 1 : 0);
    return hipGetLastError();
}

// token tiles of 64 when they fill the chip, of 32 for the few-thousand-row products
template <int K, bool RELU>
hipError_t launch_ws_any(const void *x, const void *w, const void *bias, void *y, int64_t T, int N, int64_t ldx, int64_t ldy,
                         hipStream_t st)
{
    const int ny = (N + kWavesW * 32 - 1) / (kWavesW * 32);
    const char *ys = getenv("MDETR_TOKEN_GEMM_YSTAGE");          // 0: 8-byte pieces straight from the accumulators (A/B runs)
    const char *tt = getenv("MDETR_TOKEN_GEMM_WS_TT");           // 32 | 64: force the token tile (A/B runs)
    const bool big = tt ? atoi(tt) == 64 : (T + 63) / 64 * ny >= 256;
    if (ys && atoi(ys) == 0) {
        if (big) return launch_ws<K, 64, RELU, false>(x, w, bias, y, T, N, ldx, ldy, st);
        return launch_ws<K, 32, RELU, false>(x, w, bias, y, T, N, ldx, ldy, st);
    }
    if (big) return launch_ws<K, 64, RELU, true>(x, w, bias, y, T, N, ldx, ldy, st);
    return launch_ws<K, 32, RELU, true>(x, w, bias, y, T, N, ldx, ldy, st);
}

template <int K, int NB, bool RELU>
hipError_t launch_direct(const void *x, const void *w, const void *bias, void *y, int64_t T, int N, int64_t ldx, int64_t ldy,
                         hipStream_t st)
{
    constexpr size_t lds = static_cast<size_t>(NB) * 32 * (K + 8) * 2 + NB * 32 * 4;
    static_assert(lds <= 160 * 1024, "weight block does not fit the LDS");
    auto kern = token_gemm_direct_kernel<K, NB, RELU>;
    static bool attr_set[64] = {};
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = -1;
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    const int64_t tiles = (T + 31) / 32;
    int64_t gx = (tiles + kWavesG - 1) / kWavesG;
    const int ny = (N + NB * 32 - 1) / (NB * 32);
    int per_cu = static_cast<int>((160 * 1024) / lds);
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);         // (two waves per SIMD is what the register budget allows)
    const int64_t cap = (256 * per_cu) / ny > 0 ? (256 * per_cu) / ny : 1;
    if (gx > cap) gx = cap;
    gx = (gx + 7) / 8 * 8;                                       // whole rounds over the XCDs (idle workgroups leave at once)
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(gx * ny)), dim3(kWavesG * 64), lds, st,
                       static_cast<const __bf16 *>(x), static_cast<const __bf16 *>(w), static_cast<const __bf16 *>(bias),
                       static_cast<__bf16 *>(y), T, N, ldx, ldy, static_cast<int>(gx), ny);
    return hipGetLastError();
}

template <int K, int NB, bool RELU>
hipError_t launch(const void *x, const void *w, const void *bias, void *y, int64_t T, int N, int64_t ldx, int64_t ldy,
                  hipStream_t st)
{
    constexpr size_t lds = static_cast<size_t>(NB) * 32 * (K + 8) * 2 + kWavesG * 32 * kSlabPad * 2 + NB * 32 * 4;
    static_assert(lds <= 160 * 1024, "weight block does not fit the LDS");
    auto kern = token_gemm_kernel<K, NB, RELU>;
    static bool attr_set[64] = {};                           // the attribute is per device: one process may drive several GPUs
    int dev_ = 0;
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = -1;
    if (dev_ < 0 || dev_ >= 64 || !attr_set[dev_]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return e;
        if (dev_ >= 0 && dev_ < 64) attr_set[dev_] = true;
    }
    const int64_t tiles = (T + 31) / 32;
    int64_t gx = (tiles + kWavesG - 1) / kWavesG;
    const int ny = (N + NB * 32 - 1) / (NB * 32);
    // as many workgroups as the CUs can hold at once (LDS decides: a 135 KB weight block = one per CU, a 34 KB one = three)
    const int per_cu = static_cast<int>((160 * 1024) / lds) > 0 ? static_cast<int>((160 * 1024) / lds) : 1;
    const int64_t cap = (256 * per_cu) / ny > 0 ? (256 * per_cu) / ny : 1;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(gx), static_cast<unsigned>(ny)), dim3(kWavesG * 64), lds, st,
                       static_cast<const __bf16 *>(x), static_cast<const __bf16 *>(w), static_cast<const __bf16 *>(bias),
                       static_cast<__bf16 *>(y), T, N, ldx, ldy);
    return hipGetLastError();
}

}  // namespace

bool token_gemm_supported(int64_t T, int N, int K, int64_t ldx, int64_t ldy, const void *x, const void *w, const void *y)
{
    const auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return T > 0 && (K == 512 || K == 256 || K == 128 || K == 64) && N > 0 && N % 8 == 0 && ldx % 8 == 0 && ldy % 4 == 0 && ldx >= K && ldy >= N &&
           al(x) && al(w) && (reinterpret_cast<uintptr_t>(y) & 7) == 0;
}

hipError_t token_gemm_launch(const void *x, const void *w, const void *bias, void *y, int64_t T, int N, int K,
                             int64_t ldx, int64_t ldy, bool relu, hipStream_t st)
{
    const char *dv_ = getenv("MDETR_TOKEN_GEMM_DIRECT");
    const bool direct = dv_ && atoi(dv_) != 0;
    if (dv_ && atoi(dv_) == 2 && K == 512)                      // 32 KB of weight per wave: 32-token tiles only (100 KB of LDS)
        return relu ? launch_ws<512, 32, true, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch_ws<512, 32, false, true>(x, w, bias, y, T, N, ldx, ldy, st);
    if (dv_ && atoi(dv_) == 2 && (K == 256 || K == 128 || K == 64)) {                 // the weight-in-registers form
        if (K == 256) return relu ? launch_ws_any<256, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch_ws_any<256, false>(x, w, bias, y, T, N, ldx, ldy, st);
        if (K == 128) return relu ? launch_ws_any<128, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch_ws_any<128, false>(x, w, bias, y, T, N, ldx, ldy, st);
        return relu ? launch_ws_any<64, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch_ws_any<64, false>(x, w, bias, y, T, N, ldx, ldy, st);
    }
    if (direct && K == 512) return relu ? launch_direct<512, 4, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch_direct<512, 4, false>(x, w, bias, y, T, N, ldx, ldy, st);
    if (direct && K == 128) return relu ? launch_direct<128, 8, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch_direct<128, 8, false>(x, w, bias, y, T, N, ldx, ldy, st);
    if (direct && K == 64) {
        if (N <= 64) return relu ? launch_direct<64, 2, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch_direct<64, 2, false>(x, w, bias, y, T, N, ldx, ldy, st);
        return relu ? launch_direct<64, 8, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch_direct<64, 8, false>(x, w, bias, y, T, N, ldx, ldy, st);
    }
    if (K == 512)          // 128 output features per workgroup (133 KiB of weight); wider N re-reads x per block
        return relu ? launch<512, 4, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch<512, 4, false>(x, w, bias, y, T, N, ldx, ldy, st);
    if (K == 256) {
        if (direct) {                                                        // the direct form: 64 or 128 output features per workgroup
            if (N <= 64) return relu ? launch_direct<256, 2, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch_direct<256, 2, false>(x, w, bias, y, T, N, ldx, ldy, st);
            return relu ? launch_direct<256, 4, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch_direct<256, 4, false>(x, w, bias, y, T, N, ldx, ldy, st);
        }
        int nb = N <= 64 ? 2 : (N <= 128 ? 4 : 8);                            // (a 64-wide output fits two blocks: 34 KB of weight, three workgroups per CU)
        if (const char *ev = getenv("MDETR_TOKEN_GEMM_NB")) {                 // A/B runs: output blocks of 32 per workgroup
            const int f = atoi(ev);
            if (f == 2 || f == 4 || f == 8) nb = f;
        }
        if (nb == 2) return relu ? launch<256, 2, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch<256, 2, false>(x, w, bias, y, T, N, ldx, ldy, st);
        if (nb == 4) return relu ? launch<256, 4, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch<256, 4, false>(x, w, bias, y, T, N, ldx, ldy, st);
        return relu ? launch<256, 8, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch<256, 8, false>(x, w, bias, y, T, N, ldx, ldy, st);
    }
    if (K == 64)           // the backbone's 64 -> 256 expansions (245 760 tokens in layer1): one slab per tile
        return relu ? launch<64, 8, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch<64, 8, false>(x, w, bias, y, T, N, ldx, ldy, st);
    return relu ? launch<128, 8, true>(x, w, bias, y, T, N, ldx, ldy, st) : launch<128, 8, false>(x, w, bias, y, T, N, ldx, ldy, st);
}

}  // namespace mdetr
