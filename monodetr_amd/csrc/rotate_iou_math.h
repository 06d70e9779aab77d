// monodetr_amd/csrc/rotate_iou_math.h -- overlap of two rotated boxes (x, y, dx, dy, angle) as the reference's KITTI
// evaluation computes it (lib/datasets/kitti/kitti_eval_python/rotate_iou.py:17-258, numba-CUDA device functions):
// corners of each box inside the other + pairwise edge crossings, ordered around their centroid by a monotone angle
// key with an insertion sort, area by a triangle fan.  Shared by the HIP kernel (rotate_iou.hip) and the g++ host
// build of the CPU tests.  float32 with one rounding per operation (contraction off) -- the arithmetic the oracle's
// fixture was recorded with (oracle/kitti_eval.py); cos / sin are the correctly rounded float32 values.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define MDETR_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define MDETR_HD inline
#endif

namespace mdetr {

MDETR_HD void riou_corners(const float *box, float *c)            // rbbox_to_corners (:206-229)
{
#pragma clang fp contract(off)
    const float ca = static_cast<float>(cos(static_cast<double>(box[4]))), sa = static_cast<float>(sin(static_cast<double>(box[4])));
    const float hx = box[2] / 2.0f, hy = box[3] / 2.0f;
    const float xs[4] = {-hx, -hx, hx, hx}, ys[4] = {-hy, hy, hy, -hy};
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = ca * xs[i] + sa * ys[i] + box[0];
        c[2 * i + 1] = -sa * xs[i] + ca * ys[i] + box[1];
    }
}

MDETR_HD bool riou_inside(float px, float py, const float *q)     // point_in_quadrilateral (:166-182)
{
#pragma clang fp contract(off)
    const float ab0 = q[2] - q[0], ab1 = q[3] - q[1], ad0 = q[6] - q[0], ad1 = q[7] - q[1];
    const float ap0 = px - q[0], ap1 = py - q[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0.f && adad >= adap && adap >= 0.f;
}

MDETR_HD bool riou_crossing(const float *p1, const float *p2, int i, int j, float *out)   // line_segment_intersection (:79-122)
{
#pragma clang fp contract(off)
    const float ax = p1[2 * i], ay = p1[2 * i + 1], bx = p1[2 * ((i + 1) & 3)], by = p1[2 * ((i + 1) & 3) + 1];
    const float cx = p2[2 * j], cy = p2[2 * j + 1], dx = p2[2 * ((j + 1) & 3)], dy = p2[2 * ((j + 1) & 3) + 1];
    const float ba0 = bx - ax, ba1 = by - ay, da0 = dx - ax, ca0 = cx - ax, da1 = dy - ay, ca1 = cy - ay;
    const bool acd = da1 * ca0 > ca1 * da0;
    const bool bcd = (dy - by) * (cx - bx) > (cy - by) * (dx - bx);
    if (acd == bcd) return false;
    const bool abc = ca1 * ba0 > ba1 * ca0;
    const bool abd = da1 * ba0 > ba1 * da0;
    if (abc == abd) return false;
    const float dc0 = dx - cx, dc1 = dy - cy;
    const float abba = ax * by - bx * ay, cddc = cx * dy - dx * cy;
    const float dh = ba1 * dc0 - ba0 * dc1;
    out[0] = (abba * dc0 - ba0 * cddc) / dh;
    out[1] = (abba * dc1 - ba1 * cddc) / dh;
    return true;
}

constexpr int kRiouMaxPts = 24;      // 8 corner hits + 16 crossings (the reference's 8-point buffer overflows beyond 8; see rotate_iou.hip)

// intersection area of two boxes (inter, :232-246)
MDETR_HD float riou_intersection(const float *b1, const float *b2)
{
#pragma clang fp contract(off)
    float p1[8], p2[8], px[kRiouMaxPts], py[kRiouMaxPts], key[kRiouMaxPts];
    riou_corners(b1, p1);
    riou_corners(b2, p2);
    int n = 0;
    for (int i = 0; i < 4; ++i) {                                  // quadrilateral_intersection (:185-203)
        if (riou_inside(p1[2 * i], p1[2 * i + 1], p2)) { px[n] = p1[2 * i]; py[n] = p1[2 * i + 1]; ++n; }
        if (riou_inside(p2[2 * i], p2[2 * i + 1], p1)) { px[n] = p2[2 * i]; py[n] = p2[2 * i + 1]; ++n; }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float t[2];
            if (riou_crossing(p1, p2, i, j, t)) { px[n] = t[0]; py[n] = t[1]; ++n; }
        }
    if (n == 0) return 0.f;
    float cx = 0.f, cy = 0.f;                                      // sort_vertex_in_convex_polygon (:33-72)
    for (int i = 0; i < n; ++i) { cx = cx + px[i]; cy = cy + py[i]; }
    cx = cx / static_cast<float>(n);
    cy = cy / static_cast<float>(n);
    for (int i = 0; i < n; ++i) {
        float vx = px[i] - cx, vy = py[i] - cy;
        const float d = static_cast<float>(sqrt(static_cast<double>(vx * vx + vy * vy)));
        vx = vx / d;
        vy = vy / d;
        key[i] = vy < 0.f ? -2.f - vx : vx;
    }
    for (int i = 1; i < n; ++i) {
        if (key[i - 1] > key[i]) {
            const float k = key[i], tx = px[i], ty = py[i];
            int j = i;
            while (j > 0 && key[j - 1] > k) { key[j] = key[j - 1]; px[j] = px[j - 1]; py[j] = py[j - 1]; --j; }
            key[j] = k; px[j] = tx; py[j] = ty;
        }
    }
    float total = 0.f;                                             // area (:23-30)
    for (int i = 0; i + 2 < n; ++i) {
        const float t = ((px[0] - px[i + 2]) * (py[i + 1] - py[i + 2]) - (py[0] - py[i + 2]) * (px[i + 1] - px[i + 2])) / 2.0f;
        total = total + fabsf(t);
    }
    return total;
}

// devRotateIoUEval (:249-258) with the kernel's argument order (:293-296): rbox1 = query box, rbox2 = box
MDETR_HD float riou_pair(const float *qbox, const float *box, int criterion)
{
#pragma clang fp contract(off)
    const float a1 = qbox[2] * qbox[3], a2 = box[2] * box[3];
    const float ai = riou_intersection(qbox, box);
    if (criterion == -1) return ai / (a1 + a2 - ai);
    if (criterion == 0) return ai / a1;
    if (criterion == 1) return ai / a2;
    return ai;
}

// d3_box_overlap_kernel (eval.py:195-221) for one pair: float64, camera frame (y down, origin at the box bottom);
// boxes are (x, y, z, l, h, w, ry).  `bev` = intersection area of the ground-plane rectangles (criterion 2 above).
MDETR_HD double box3d_overlap(const double *b, const double *q, double bev, int criterion)
{
#pragma clang fp contract(off)
    if (!(bev > 0.0)) return bev;
    const double top = b[1] < q[1] ? b[1] : q[1];
    const double lo_b = b[1] - b[4], lo_q = q[1] - q[4];
    const double ih = top - (lo_b > lo_q ? lo_b : lo_q);
    if (!(ih > 0.0)) return 0.0;
    const double v1 = b[3] * b[4] * b[5], v2 = q[3] * q[4] * q[5];
    const double inc = ih * bev;
    const double ua = criterion == -1 ? (v1 + v2 - inc) : (criterion == 0 ? v1 : (criterion == 1 ? v2 : inc));
    return inc / ua;
}

}  // namespace mdetr
