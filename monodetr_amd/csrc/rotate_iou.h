// monodetr_amd/csrc/rotate_iou.h -- internal launcher declarations (see rotate_iou.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdetr {

// Segmented all-pairs overlaps: frame f owns boxes [box_start[f], box_start[f+1]) and query boxes
// [qbox_start[f], qbox_start[f+1]); its row-major [n_f, k_f] block of `out` starts at out_start[f].
hipError_t rotate_iou_launch(const float *boxes, const float *qboxes, const int64_t *box_start, const int64_t *qbox_start,
                             const int64_t *out_start, int n_frames, int64_t total_pairs, int criterion, float *out,
                             hipStream_t st);
hipError_t box3d_overlap_launch(const double *boxes, const double *qboxes, const int64_t *box_start, const int64_t *qbox_start,
                                const int64_t *out_start, int n_frames, int64_t total_pairs, int criterion, double *out,
                                hipStream_t st);

}  // namespace mdetr
