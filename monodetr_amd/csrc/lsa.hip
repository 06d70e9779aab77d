// monodetr_amd/csrc/lsa.hip -- batched linear sum assignment (Hungarian matching) on the GPU.
//
// The reference matches predictions to ground truth on the HOST: it copies the cost matrix to the
// CPU and calls scipy.optimize.linear_sum_assignment once per (decoder layer, image, query group)
// -- 3 x 8 x 11 = 264 calls and a device->host sync per training iteration
// (lib/models/monodetr/matcher.py:87-103).  Here one wave64 solves one assignment problem with the
// shortest-augmenting-path (Jonker-Volgenant / "Hungarian with potentials") algorithm:
//   * rows = the k <= 64 ground-truth objects of an image, columns = the n <= 128 queries of a group;
//     lane j owns column j -- and column j + 64 when a group has more than 64 queries -- (potential v_j, slack
//     minv_j, predecessor way_j, assigned row p_j);
//   * the per-step argmin over the free columns is a 6-step butterfly over the wave;
//   * all arithmetic in float64, as scipy does on the same fp32 costs, so the optimum is the same
//     (when the optimum is not unique either solver may return any optimal assignment).
// Problems are tiny (<= 50 x 50); all of them run concurrently, no host involvement, graph-capturable.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mdetr_wave.h>

#include "lsa.h"
#include "pair_losses_math.h"

namespace mdetr {
namespace {

constexpr int kMaxDim = 64;
constexpr int kWavesPerBlock = 4;

struct MinLoc { double v; int j; };

__device__ __forceinline__ MinLoc wave_argmin(double v, int j)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const int oj = __shfl_xor(j, o);
        if (ov < v || (ov == v && oj < j)) { v = ov; j = oj; }
    }
    return {v, j};
}

// inputs of the fused form: the cost is evaluated in the kernel from the predictions and the padded ground
// truth (pl_match_cost, pair_losses_math.h) instead of being read from a cost matrix built by ~45
// framework kernels
struct LsaFused {
    const float *logits, *boxes;       // [L*B, Q, C], [L*B, Q, 6]
    const long long *labels;           // [B, kmax]
    const float *boxes3d;              // [B, kmax, 6]
    int num_classes;
    MatchWeights w;
};

// cost(problem, target t, query column j) = C[base + j * q_stride + t * t_stride]   (FUSED: computed)
// CPL columns per lane: lane owns columns lane, lane + 64, ... (n <= 64 CPL; the 100 queries per group of the
// 512 x 1760 configuration need CPL = 2).  Column indices and everything derived from the wave-wide argmin are
// wave-uniform, so "the value column j holds" is a uniform choice of the register followed by a lane read.
template <bool FUSED, int CPL>
__global__ __launch_bounds__(kWavesPerBlock * 64)
void lsa_kernel(const float *__restrict__ C, const int *__restrict__ num_targets, int *__restrict__ assign,
                int num_problems, int groups, int n, int kmax,
                int64_t img_stride, int64_t q_stride, int64_t t_stride, int images_per_layer, const LsaFused fz)
{
    constexpr int kCols = 64 * CPL, kStride = kCols + 1;
    // per wave: costs [kmax][kCols + 1] fp32 (exact in fp64 on read) + row potentials u[64] fp64
    MDETR_DYNAMIC_LDS(unsigned char, lsa_smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int prob = blockIdx.x * kWavesPerBlock + wave;
    if (prob >= num_problems) return;                                 // wave-uniform
    // problem -> (layer*image, group)
    const int g = prob % groups, li = prob / groups;                  // li = layer * images_per_layer + image
    const int image = li % images_per_layer;
    int k = num_targets[image];
    k = k < 0 ? 0 : (k > kmax ? kmax : k);
    int *out = assign + static_cast<int64_t>(prob) * kmax;
    for (int t = lane; t < kmax; t += 64) out[t] = -1;
    if (k == 0) return;

    const size_t per_wave = static_cast<size_t>(kmax) * kStride * sizeof(float) + kMaxDim * sizeof(double);
    double *u = reinterpret_cast<double *>(lsa_smem + wave * per_wave);
    float (*a)[kStride] = reinterpret_cast<float (*)[kStride]>(lsa_smem + wave * per_wave + kMaxDim * sizeof(double));
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int colj = lane + 64 * c;
        if (FUSED) {
            const int64_t row = (static_cast<int64_t>(li) * groups + g) * n + (colj < n ? colj : 0);   // (l, b, query)
            const float *lg = fz.logits + row * fz.num_classes, *bx = fz.boxes + row * 6;
            for (int t = 0; t < k; ++t) {
                const int64_t tk = static_cast<int64_t>(image) * kmax + t;
                a[t][colj] = colj < n ? pl_match_cost(lg, bx, static_cast<int>(fz.labels[tk]), fz.boxes3d + tk * 6, fz.w) : 0.f;
            }
        } else {
            const float *Cp = C + static_cast<int64_t>(li) * img_stride + static_cast<int64_t>(g) * n * q_stride;
            for (int t = 0; t < k; ++t)
                a[t][colj] = colj < n ? Cp[colj * q_stride + t * t_stride] : 0.f;
        }
    }
    if (lane < kMaxDim) u[lane] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const double INF = 1e300;
    bool col[CPL];
    double v[CPL];
    int p[CPL];                                                       // row assigned to each of this lane's columns
#pragma unroll
    for (int c = 0; c < CPL; ++c) { col[c] = lane + 64 * c < n; v[c] = 0.0; p[c] = -1; }
    // p of column j (j wave-uniform, >= 0)
    auto p_of = [&](int j) __attribute__((always_inline)) {
        int mine = p[0];
#pragma unroll
        for (int c = 1; c < CPL; ++c) mine = (j >> 6) == c ? p[c] : mine;
        return __shfl(mine, j & 63);
    };
    for (int i = 0; i < k; ++i) {
        double minv[CPL];
        int way[CPL];                                                 // -1 = reached from the dummy column
        bool used[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) { minv[c] = INF; way[c] = -2; used[c] = false; }
        int j0 = -1;                                                  // current column (-1 = dummy column holding row i)
        // Every pass marks one more column as used, so a free column is reached within i + 1 <= n passes -- for
        // finite costs.  With NaN / inf costs (diverged predictions) the candidates stop being ordered: the argmin can
        // return a used column again or differ between lanes, and nothing then guarantees that an unbounded search
        // ends -- on the GPU that is a hung queue, not a wrong number.  Both loops of a row are therefore bounded by
        // n + 1 trips; a row whose search does not reach a free column stays unmatched (the result is meaningless
        // for such input, but the kernel returns and the output is still a matching).
        bool reached = false;
        for (int pass = 0; pass <= n; ++pass) {
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (lane + 64 * c == j0) used[c] = true;
            const int i0 = j0 < 0 ? i : p_of(j0);
            const double ui0 = u[i0];
            double cand = INF;
            int cj = lane;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                if (col[c] && !used[c]) {
                    const double cur = static_cast<double>(a[i0][lane + 64 * c]) - ui0 - v[c];
                    if (cur < minv[c]) { minv[c] = cur; way[c] = j0; }
                    if (minv[c] < cand) { cand = minv[c]; cj = lane + 64 * c; }      // (the lower column wins a tie)
                }
            }
            const MinLoc m = wave_argmin(cand, cj);
            const double delta = m.v;
            // potentials: rows of used columns (distinct rows) and the dummy column's row i
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                if (used[c]) { u[p[c]] += delta; v[c] -= delta; }
                else minv[c] -= delta;
            }
            if (lane == 0) u[i] += delta;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            j0 = m.j;
            if (p_of(j0) < 0) { reached = true; break; }              // free column reached
        }
        if (!reached) continue;                                       // wave-uniform (j0 and p[j0] are)
        // augment along the alternating path back to the dummy column
        for (int hop = 0; hop <= n && j0 >= 0; ++hop) {
            int mine = way[0];
#pragma unroll
            for (int c = 1; c < CPL; ++c) mine = (j0 >> 6) == c ? way[c] : mine;
            const int j1 = __shfl(mine, j0 & 63);
            const int pj1 = j1 < 0 ? i : p_of(j1);
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (lane + 64 * c == j0) p[c] = pj1;
            j0 = j1;
        }
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c)
        if (col[c] && p[c] >= 0) out[p[c]] = g * n + lane + 64 * c;
}

size_t lsa_lds_bytes(int kmax, int cpl)
{
    return kWavesPerBlock * (static_cast<size_t>(kmax) * (64 * cpl + 1) * sizeof(float) + kMaxDim * sizeof(double));
}

}  // namespace

hipError_t lsa_launch(const float *cost, const int *num_targets, int *assign, int layers, int images, int groups,
                      int n, int kmax, int64_t img_stride, int64_t q_stride, int64_t t_stride, hipStream_t st)
{
    const int num_problems = layers * images * groups;
    if (num_problems == 0 || kmax == 0) return hipSuccess;
    const dim3 grid((num_problems + kWavesPerBlock - 1) / kWavesPerBlock), block(kWavesPerBlock * 64);
    if (n > 64) {
        static bool attr_set[64] = {};
        int dev = 0;
        hipError_t err = hipGetDevice(&dev);
        if (err != hipSuccess) return err;
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            if ((err = hipFuncSetAttribute(reinterpret_cast<const void *>(lsa_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return err;
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
        hipLaunchKernelGGL((lsa_kernel<false, 2>), grid, block, lsa_lds_bytes(kmax, 2), st, cost, num_targets, assign, num_problems, groups, n, kmax,
                           img_stride, q_stride, t_stride, images, LsaFused{});
    } else {
        hipLaunchKernelGGL((lsa_kernel<false, 1>), grid, block, lsa_lds_bytes(kmax, 1), st, cost, num_targets, assign, num_problems, groups, n, kmax,
                           img_stride, q_stride, t_stride, images, LsaFused{});
    }
    return hipGetLastError();
}

hipError_t lsa_fused_launch(const float *logits, const float *boxes, const int64_t *labels, const float *boxes3d,
                            const int *num_targets, int *assign, int layers, int images, int groups, int n, int kmax,
                            int num_classes, float w_class, float w_bbox, float w_center, float w_giou, float alpha,
                            hipStream_t st)
{
    const int num_problems = layers * images * groups;
    if (num_problems == 0 || kmax == 0) return hipSuccess;
    const dim3 grid((num_problems + kWavesPerBlock - 1) / kWavesPerBlock), block(kWavesPerBlock * 64);
    const LsaFused fz{logits, boxes, reinterpret_cast<const long long *>(labels), boxes3d, num_classes,
                      MatchWeights{w_class, w_bbox, w_center, w_giou, alpha}};
    if (n > 64) {
        static bool attr_set[64] = {};
        int dev = 0;
        hipError_t err = hipGetDevice(&dev);
        if (err != hipSuccess) return err;
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            if ((err = hipFuncSetAttribute(reinterpret_cast<const void *>(lsa_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return err;
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
        hipLaunchKernelGGL((lsa_kernel<true, 2>), grid, block, lsa_lds_bytes(kmax, 2), st, nullptr, num_targets, assign, num_problems, groups, n, kmax,
                           0, 0, 0, images, fz);
    } else {
        hipLaunchKernelGGL((lsa_kernel<true, 1>), grid, block, lsa_lds_bytes(kmax, 1), st, nullptr, num_targets, assign, num_problems, groups, n, kmax,
                           0, 0, 0, images, fz);
    }
    return hipGetLastError();
}

}  // namespace mdetr
