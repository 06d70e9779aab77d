// monodetr_amd/csrc/pair_losses.hip -- MonoDETR's matched-pair losses for all decoder levels in ONE
// launch, and their gradients in one more.
//
// After the matching moved to the device the criterion is ~230 small framework kernels forward and as
// many backward on tensors of a few thousand elements (13 200 query rows): pure launch overhead on a
// step that is launch-bound.  Everything except the depth-map loss is per-query
// arithmetic (pair_losses_math.h) plus a handful of sums, so:
//   forward   one thread per (level, image, query): find the query's matched slot, evaluate the focal
//             term of its C logits and -- if matched -- the six pair losses; wave/block reduction, one
//             fp32 atomic per block and loss row into a 12 x L workspace; the LAST block to finish
//             normalises (num_boxes, the dimension-aware factor, class error, cardinality), writes the
//             [9, L] result and clears the workspace for the next call.
//   backward  one thread per row recomputes its terms and writes its rows of the five gradient tensors
//             (zeros for unmatched queries): no zero-fill, no atomics.
// Latency-bound by construction (52 blocks); what it buys is ~450 launches per training step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pair_losses.h"

namespace mdetr {
namespace {

constexpr int kAcc = kPairLossRows + 3;          // 9 result rows + sum relative |d| + hits + matches

struct PairLossWork {                            // workspace layout
    float *sums;                                 // [L][kAcc]
    int *card;                                   // [L][B] foreground-argmax counts
    unsigned *done;                              // blocks finished
};

__device__ __forceinline__ PairLossWork carve(void *ws, int L, int B)
{
    PairLossWork w;
    w.sums = static_cast<float *>(ws);
    w.card = reinterpret_cast<int *>(w.sums + L * kAcc);
    w.done = reinterpret_cast<unsigned *>(w.card + L * B);
    return w;
}

__global__ __launch_bounds__(256)
void pair_losses_fwd_kernel(const PairLossDims d, const PairLossIn in, const int32_t *__restrict__ num,
                            float nb_host, const float *__restrict__ nb_dev, float *__restrict__ out,
                            float *__restrict__ comp, void *__restrict__ ws)
{
    __shared__ float red[4][kAcc];
    __shared__ bool last;
    const PairLossWork w = carve(ws, d.L, d.B);
    const int rows_per_level = d.B * d.Q;
    // a block never straddles levels: blocks are laid out per level
    const int blocks_per_level = (rows_per_level + 255) / 256;
    const int l = blockIdx.x / blocks_per_level;
    const int r = (blockIdx.x % blocks_per_level) * 256 + threadIdx.x;
    float acc[kAcc];
#pragma unroll
    for (int i = 0; i < kAcc; ++i) acc[i] = 0.f;
    // the cardinality count: one atomic per (wave, image) instead of one per row -- 13 200 atomics on 24 addresses queue up behind
    // each other at the L2 (they were most of this kernel's 100 us); a wave's 64 consecutive rows belong to at most two images
    bool counted = false;
    int my_b = -1;
    if (r < rows_per_level) {
        const int b = r / d.Q, q = r - b * d.Q;
        counted = pl_row_forward(d, in, l, b, q, acc);
        my_b = b;
    }
    for (unsigned long long rest = __ballot(counted); rest != 0ull;) {
        const int first = __builtin_ctzll(rest);
        const int bb = __shfl(my_b, first);
        const unsigned long long same = __ballot(counted && my_b == bb);
        if (static_cast<int>(threadIdx.x & 63) == first) atomicAdd(&w.card[l * d.B + bb], __popcll(same));
        rest &= ~same;
    }
    // wave reduction (64 lanes), then the 4 waves through LDS
#pragma unroll
    for (int i = 0; i < kAcc; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < kAcc) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (v != 0.f) atomicAdd(&w.sums[l * kAcc + threadIdx.x], v);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(w.done, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // finalise: one thread per level
    if (static_cast<int>(threadIdx.x) < d.L) {
        const int lv = threadIdx.x;
        float s[kAcc];                                                  // agent-scope loads: the sums were built by
        for (int i = 0; i < kAcc; ++i)                                  // atomics issued from other XCDs
            s[i] = __hip_atomic_load(w.sums + lv * kAcc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float nb = nb_dev ? *nb_dev : nb_host;
        const float s_rel = s[kPairLossRows], s_dim = s[kLossDim];
        const float cf = s_dim / fmaxf(s_rel, 1e-12f);
        comp[lv] = cf;
        out[kLossCe * d.L + lv] = s[kLossCe] / nb;
        out[kLossCenter * d.L + lv] = s[kLossCenter] / nb;
        out[kLossBbox * d.L + lv] = s[kLossBbox] / nb;
        out[kLossGiou * d.L + lv] = s[kLossGiou] / nb;
        out[kLossDepth * d.L + lv] = s[kLossDepth] / nb;
        out[kLossDim * d.L + lv] = s_rel * cf / nb;
        out[kLossAngle * d.L + lv] = s[kLossAngle] / nb;
        const float hits = s[kPairLossRows + 1], nmatch = s[kPairLossRows + 2];
        out[kClassError * d.L + lv] = 100.f - (nmatch > 0.f ? hits * 100.f / nmatch : 0.f);
        float ce = 0.f;
        for (int b = 0; b < d.B; ++b) {
            const int cnt = __hip_atomic_load(w.card + lv * d.B + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float diff = static_cast<float>(cnt) - static_cast<float>(num[b]);
            ce += diff < 0.f ? -diff : diff;
        }
        out[kCardinality * d.L + lv] = ce / static_cast<float>(d.B);
    }
    __syncthreads();
    // leave the workspace clean for the next launch
    for (int i = threadIdx.x; i < d.L * kAcc; i += 256) w.sums[i] = 0.f;
    for (int i = threadIdx.x; i < d.L * d.B; i += 256) w.card[i] = 0;
    if (threadIdx.x == 0) *w.done = 0u;
}

__global__ __launch_bounds__(256)
void pair_losses_bwd_kernel(const PairLossDims d, const PairLossIn in, const float *__restrict__ grad_out,
                            const float *__restrict__ comp, float nb_host, const float *__restrict__ nb_dev,
                            float *__restrict__ g_logits, float *__restrict__ g_boxes, float *__restrict__ g_dims,
                            float *__restrict__ g_depths, float *__restrict__ g_angles)
{
    const int64_t row = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t total = static_cast<int64_t>(d.L) * d.B * d.Q;
    if (row >= total) return;
    const int l = static_cast<int>(row / (static_cast<int64_t>(d.B) * d.Q));
    const int rem = static_cast<int>(row - static_cast<int64_t>(l) * d.B * d.Q);
    const int b = rem / d.Q, q = rem - b * d.Q;
    const float inv = 1.f / (nb_dev ? *nb_dev : nb_host);
    float w[kNumWeighted];
#pragma unroll
    for (int i = 0; i < kNumWeighted; ++i) w[i] = grad_out[i * d.L + l] * inv;
    pl_row_backward(d, in, l, b, q, w, comp[l], g_logits, g_boxes, g_dims, g_depths, g_angles);
}

}  // namespace

int64_t pair_losses_workspace_bytes(int L, int B)
{
    return static_cast<int64_t>(L) * kAcc * sizeof(float) + static_cast<int64_t>(L) * B * sizeof(int) + 16;
}

hipError_t pair_losses_forward_launch(const PairLossDims &d, const PairLossIn &in, const int32_t *num,
                                      float num_boxes, const float *num_boxes_dev, float *out, float *comp,
                                      void *workspace, hipStream_t st)
{
    const int blocks_per_level = (d.B * d.Q + 255) / 256;
    hipLaunchKernelGGL(pair_losses_fwd_kernel, dim3(static_cast<unsigned>(d.L * blocks_per_level)), dim3(256), 0, st,
                       d, in, num, num_boxes, num_boxes_dev, out, comp, workspace);
    return hipGetLastError();
}

hipError_t pair_losses_backward_launch(const PairLossDims &d, const PairLossIn &in, const float *grad_out,
                                       const float *comp, float num_boxes, const float *num_boxes_dev,
                                       float *g_logits, float *g_boxes, float *g_dims, float *g_depths,
                                       float *g_angles, hipStream_t st)
{
    const int64_t total = static_cast<int64_t>(d.L) * d.B * d.Q;
    hipLaunchKernelGGL(pair_losses_bwd_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st,
                       d, in, grad_out, comp, num_boxes, num_boxes_dev, g_logits, g_boxes, g_dims, g_depths, g_angles);
    return hipGetLastError();
}

}  // namespace mdetr
