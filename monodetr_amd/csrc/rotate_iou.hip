// monodetr_amd/csrc/rotate_iou.hip -- rotated-box overlaps of the KITTI evaluation (bird's-eye-view IoU and 3-D IoU).
//
// Reference: rotate_iou_gpu_eval (lib/datasets/kitti/kitti_eval_python/rotate_iou.py:262-330), a numba-CUDA kernel
// with 64-thread blocks (one wavefront) where a thread owns one box and loops over 64 query boxes staged in shared
// memory -- called per PART of 50 frames on ALL box x query pairs of the part (eval.py:404-486), of which only the
// per-frame diagonal blocks are used afterwards (1/50 of the pairs); the 3-D overlap then multiplies the BEV
// intersection by the height overlap in a second, CPU-side pass (eval.py:195-228).
//
// Here: one thread per USED pair.  The launch is segmented by frame (prefix arrays of box / query / output offsets,
// a binary search per thread), so only within-frame pairs are computed, for the whole split in one launch, and the
// 3-D variant finishes the pair in the same thread.  The work is ALU-bound and branchy (~1-2 kflop per overlapping
// pair, early exit for disjoint ones); the boxes (20 / 56 bytes) are L2-resident.  Algorithmic bytes = 4 (or 8)
// per pair written + the boxes once.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rotate_iou.h"
#include "rotate_iou_math.h"

namespace mdetr {
namespace {

constexpr int kThreads = 256;

// frame of flat pair index p: largest f with out_start[f] <= p  (out_start[n_frames] = total)
__device__ __forceinline__ int frame_of(const int64_t *__restrict__ out_start, int n_frames, int64_t p)
{
    int lo = 0, hi = n_frames;                      // invariant: out_start[lo] <= p < out_start[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (out_start[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kThreads)
void rotate_iou_kernel(const float *__restrict__ boxes, const float *__restrict__ qboxes,
                       const int64_t *__restrict__ box_start, const int64_t *__restrict__ qbox_start,
                       const int64_t *__restrict__ out_start, int n_frames, int64_t total, int criterion,
                       float *__restrict__ out)
{
    const int64_t p = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (p >= total) return;
    const int f = frame_of(out_start, n_frames, p);
    const int64_t k_f = qbox_start[f + 1] - qbox_start[f], local = p - out_start[f];
    const int64_t n = local / k_f, k = local - n * k_f;
    float b[5], q[5];
    for (int i = 0; i < 5; ++i) { b[i] = boxes[(box_start[f] + n) * 5 + i]; q[i] = qboxes[(qbox_start[f] + k) * 5 + i]; }
    out[p] = riou_pair(q, b, criterion);
}

__global__ __launch_bounds__(kThreads)
void box3d_overlap_kernel(const double *__restrict__ boxes, const double *__restrict__ qboxes,
                          const int64_t *__restrict__ box_start, const int64_t *__restrict__ qbox_start,
                          const int64_t *__restrict__ out_start, int n_frames, int64_t total, int criterion,
                          double *__restrict__ out)
{
    const int64_t p = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
    if (p >= total) return;
    const int f = frame_of(out_start, n_frames, p);
    const int64_t k_f = qbox_start[f + 1] - qbox_start[f], local = p - out_start[f];
    const int64_t n = local / k_f, k = local - n * k_f;
    double b[7], q[7];
    for (int i = 0; i < 7; ++i) { b[i] = boxes[(box_start[f] + n) * 7 + i]; q[i] = qboxes[(qbox_start[f] + k) * 7 + i]; }
    const float b5[5] = {static_cast<float>(b[0]), static_cast<float>(b[2]), static_cast<float>(b[3]), static_cast<float>(b[5]), static_cast<float>(b[6])};
    const float q5[5] = {static_cast<float>(q[0]), static_cast<float>(q[2]), static_cast<float>(q[3]), static_cast<float>(q[5]), static_cast<float>(q[6])};
    const double bev = static_cast<double>(riou_pair(q5, b5, 2));          // ground-plane intersection area, float32 as the reference
    out[p] = box3d_overlap(b, q, bev, criterion);
}

}  // namespace

hipError_t rotate_iou_launch(const float *boxes, const float *qboxes, const int64_t *box_start, const int64_t *qbox_start,
                             const int64_t *out_start, int n_frames, int64_t total_pairs, int criterion, float *out,
                             hipStream_t st)
{
    if (total_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(rotate_iou_kernel, dim3(static_cast<unsigned>((total_pairs + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                       boxes, qboxes, box_start, qbox_start, out_start, n_frames, total_pairs, criterion, out);
    return hipGetLastError();
}

hipError_t box3d_overlap_launch(const double *boxes, const double *qboxes, const int64_t *box_start, const int64_t *qbox_start,
                                const int64_t *out_start, int n_frames, int64_t total_pairs, int criterion, double *out,
                                hipStream_t st)
{
    if (total_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(box3d_overlap_kernel, dim3(static_cast<unsigned>((total_pairs + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
                       boxes, qboxes, box_start, qbox_start, out_start, n_frames, total_pairs, criterion, out);
    return hipGetLastError();
}

}  // namespace mdetr
