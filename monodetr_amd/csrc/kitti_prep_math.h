// monodetr_amd/csrc/kitti_prep_math.h -- one output pixel of the reference's training image path
// (lib/datasets/kitti/kitti_dataset.py:127-163), shared by the HIP kernel (kitti_prep.hip) and by the host build
// the CPU tests compile with g++ (tests/native/host_kernels.cpp):
//
//   photometric distortion on the decoded RGB8 image, float32, result cast back to uint8   (pd.py:376-398, :138-139)
//   horizontal flip                                                                        (:141-143)
//   PIL Image.transform(AFFINE, BILINEAR): float64 interpolation, truncated to uint8       (:155-158)
//   / 255, (x - mean) / std, HWC -> CHW                                                    (:161-163)
//
// The reference materialises three intermediate images; here one output pixel is computed from its (at most) four
// source pixels directly: the distortion is a per-pixel function, the flip an index reflection.  The result is
// BIT-IDENTICAL to the reference chain, which fixes the arithmetic: float32 for the distortion evaluated operation by
// operation (numpy has no fused multiply-add -- contraction is switched off below), float64 for the interpolation
// exactly as libImaging's bilinear filter writes it, and the wrap-around uint8 cast numpy performs on x86-64.
// OpenCV's float HSV conversions are restated from its scalar code path (see oracle/kitti_pipeline.py's header for
// what that does and does not pin).
#pragma once

#include <stdint.h>

#include "../../include/monodetr_amd.h"

#if defined(__HIPCC__)
#define MDETR_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define MDETR_HD inline
#endif

namespace mdetr {

constexpr float kFltEps = 1.1920929e-07f;

MDETR_HD float kp_max3(float a, float b, float c) { const float m = a > b ? a : b; return m > c ? m : c; }
MDETR_HD float kp_min3(float a, float b, float c) { const float m = a < b ? a : b; return m < c ? m : c; }

// cv2.cvtColor(float32, COLOR_BGR2HSV): channel 0 is "B" (the reference feeds RGB, so its hue is that of the
// channel-swapped colour -- kept as is).
MDETR_HD void kp_bgr2hsv(float b, float g, float r, float &h, float &s, float &v)
{
#pragma clang fp contract(off)
    v = kp_max3(r, g, b);
    const float vmin = kp_min3(r, g, b);
    const float diff = v - vmin;
    s = diff / (fabsf(v) + kFltEps);
    const float scale = 60.0f / (diff + kFltEps);
    if (v == r) h = (g - b) * scale;
    else if (v == g) h = (b - r) * scale + 120.0f;
    else h = (r - g) * scale + 240.0f;
    if (h < 0.0f) h = h + 360.0f;
}

// cv2.cvtColor(float32, COLOR_HSV2BGR)
MDETR_HD void kp_hsv2bgr(float h, float s, float v, float &b, float &g, float &r)
{
#pragma clang fp contract(off)
    if (s == 0.0f) { b = g = r = v; return; }
    float hh = h * static_cast<float>(6.0 / 360.0);
    for (int it = 0; it < 4; ++it) {
        if (hh < 0.0f) hh = hh + 6.0f;
        if (hh >= 6.0f) hh = hh - 6.0f;
    }
    int sector = static_cast<int>(floorf(hh));
    float frac = hh - static_cast<float>(sector);
    if (sector < 0 || sector >= 6) { sector = 0; frac = 0.0f; }
    const float t1 = v * (1.0f - s);
    const float t2 = v * (1.0f - s * frac);
    const float t3 = v * (1.0f - s * (1.0f - frac));
    switch (sector) {                     // {b, g, r} = tab[{1,3,0} {1,0,2} {3,0,1} {0,2,1} {0,1,3} {2,1,0}], tab = {v, t1, t2, t3}
    case 0: b = t1; g = t3; r = v; break;
    case 1: b = t1; g = v; r = t2; break;
    case 2: b = t3; g = v; r = t1; break;
    case 3: b = v; g = t2; r = t1; break;
    case 4: b = v; g = t1; r = t3; break;
    default: b = t2; g = t1; r = v; break;
    }
}

// numpy's float32 -> uint8 cast on x86-64: truncate toward zero, keep the low 8 bits
MDETR_HD uint32_t kp_wrap_u8(float x) { return static_cast<uint32_t>(static_cast<int32_t>(x)) & 0xFFu; }

// PhotometricDistort on one pixel; px = the three stored bytes in memory order, out[c] = distorted channel c
MDETR_HD void kp_distort(const MdetrKittiImage &d, const uint8_t *px, uint32_t out[3])
{
#pragma clang fp contract(off)
    if (!(d.flags & MDETR_KITTI_DISTORT)) { out[0] = px[0]; out[1] = px[1]; out[2] = px[2]; return; }
    float c0 = static_cast<float>(px[0]), c1 = static_cast<float>(px[1]), c2 = static_cast<float>(px[2]);
    if (d.flags & MDETR_KITTI_BRIGHTNESS) { c0 = c0 + d.brightness; c1 = c1 + d.brightness; c2 = c2 + d.brightness; }
    const bool contrast = (d.flags & MDETR_KITTI_CONTRAST) != 0, first = (d.flags & MDETR_KITTI_CONTRAST_FIRST) != 0;
    if (contrast && first) { c0 = c0 * d.contrast; c1 = c1 * d.contrast; c2 = c2 * d.contrast; }
    float h, s, v;
    kp_bgr2hsv(c0, c1, c2, h, s, v);
    if (d.flags & MDETR_KITTI_SATURATION) s = s * d.saturation;
    if (d.flags & MDETR_KITTI_HUE) {
        h = h + d.hue;
        if (h > 360.0f) h = h - 360.0f;
        if (h < 0.0f) h = h + 360.0f;
    }
    kp_hsv2bgr(h, s, v, c0, c1, c2);
    if (contrast && !first) { c0 = c0 * d.contrast; c1 = c1 * d.contrast; c2 = c2 * d.contrast; }
    const uint32_t q[3] = {kp_wrap_u8(c0), kp_wrap_u8(c1), kp_wrap_u8(c2)};
    out[0] = q[d.perm & 3u];
    out[1] = q[(d.perm >> 2) & 3u];
    out[2] = q[(d.perm >> 4) & 3u];
}

struct KpTaps {                  // libImaging bilinear_filter32RGB's view of one output pixel
    bool inside, row1;           // source coordinate inside the image; second row exists
    int xa, xb, ya, yb;          // clipped tap columns / rows (in the flipped image's coordinates)
    double dx, dy;
};

MDETR_HD int kp_clip(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

MDETR_HD KpTaps kp_taps(const MdetrKittiImage &d, int ox, int oy)
{
#pragma clang fp contract(off)
    KpTaps t;
    const double xc = static_cast<double>(ox) + 0.5, yc = static_cast<double>(oy) + 0.5;
    double xin = d.inv[0] * xc + d.inv[1] * yc + d.inv[2];
    double yin = d.inv[3] * xc + d.inv[4] * yc + d.inv[5];
    t.inside = xin >= 0.0 && xin < static_cast<double>(d.width) && yin >= 0.0 && yin < static_cast<double>(d.height);
    xin = xin - 0.5;
    yin = yin - 0.5;
    const double xf = floor(xin), yf = floor(yin);
    t.dx = xin - xf;
    t.dy = yin - yf;
    const int x0 = static_cast<int>(xf), y0 = static_cast<int>(yf);
    t.xa = kp_clip(x0, d.width - 1);
    t.xb = kp_clip(x0 + 1, d.width - 1);
    t.ya = kp_clip(y0, d.height - 1);
    t.row1 = y0 + 1 >= 0 && y0 + 1 < d.height;
    t.yb = kp_clip(y0 + 1, d.height - 1);
    return t;
}

// The three normalised channel values of output pixel (ox, oy).  `img` = first byte of this image's pixels.
MDETR_HD void kp_pixel(const MdetrKittiImage &d, const uint8_t *img, int ox, int oy, const float mean[3],
                       const float stdv[3], float out[3])
{
#pragma clang fp contract(off)
    const KpTaps t = kp_taps(d, ox, oy);
    uint32_t q[3] = {0u, 0u, 0u};
    if (t.inside) {
        const bool flip = (d.flags & MDETR_KITTI_FLIP) != 0;
        const int xa = flip ? d.width - 1 - t.xa : t.xa, xb = flip ? d.width - 1 - t.xb : t.xb;
        const int64_t row = static_cast<int64_t>(d.width) * 3;
        uint32_t p00[3], p01[3], p10[3], p11[3];
        kp_distort(d, img + t.ya * row + xa * 3, p00);
        kp_distort(d, img + t.ya * row + xb * 3, p01);
        if (t.row1) {
            kp_distort(d, img + t.yb * row + xa * 3, p10);
            kp_distort(d, img + t.yb * row + xb * 3, p11);
        }
        for (int c = 0; c < 3; ++c) {
            const double a = static_cast<double>(p00[c]), b = static_cast<double>(p01[c]);
            const double v1 = a + (b - a) * t.dx;
            double v2 = v1;
            if (t.row1) {
                const double e = static_cast<double>(p10[c]), f = static_cast<double>(p11[c]);
                v2 = e + (f - e) * t.dx;
            }
            const double v = v1 + (v2 - v1) * t.dy;
            q[c] = static_cast<uint32_t>(static_cast<int32_t>(v)) & 0xFFu;
        }
    }
    for (int c = 0; c < 3; ++c) out[c] = (static_cast<float>(q[c]) / 255.0f - mean[c]) / stdv[c];
}

}  // namespace mdetr
