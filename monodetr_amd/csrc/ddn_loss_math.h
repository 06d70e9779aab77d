// monodetr_amd/csrc/ddn_loss_math.h -- per-pixel arithmetic of MonoDETR's depth-map loss
// (lib/models/monodetr/depth_predictor/ddn_loss/{ddn_loss,balancer,focalloss}.py): target depth painted
// from the ground-truth 2D boxes (nearest object wins), linear-increasing depth bins, multi-class focal
// loss with the reference's +1e-6 one-hot smoothing, foreground/background weights.  Shared by the HIP
// kernels (ddn_loss.hip) and the host build of the CPU tests (tests/native/host_kernels.cpp).
//
// Work unit = one pixel (image b, row y, column x) of the [B, D+1, H, W] depth logits.
#pragma once

#if defined(__HIPCC__)
#define MDETR_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define MDETR_HD inline
#endif

namespace mdetr {

struct DdnDims {
    int B, C, H, W, K;              // C = number of depth bins + 1
    long long sb, sc, sh, sw;       // element strides of the logits (any dense layout)
    float alpha, fg_weight, bg_weight, depth_min, depth_max;
};

MDETR_HD float ddn_floor(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return floorf(x);
#else
    return std::floor(x);
#endif
}
MDETR_HD float ddn_ceil(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return ceilf(x);
#else
    return std::ceil(x);
#endif
}
MDETR_HD float ddn_sqrt(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fsqrt_rn(x);
#else
    return std::sqrt(x);
#endif
}
MDETR_HD float ddn_exp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __expf(x);
#else
    return std::exp(x);
#endif
}
MDETR_HD float ddn_log(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __logf(x);
#else
    return std::log(x);
#endif
}

// Python slice [lo:hi] on an axis of length n (balancer.py's painting loops index with raw corners:
// negative values wrap, then clip)
MDETR_HD void ddn_slice(int lo, int hi, int n, int &start, int &stop)
{
    start = lo < 0 ? (lo + n < 0 ? 0 : lo + n) : (lo > n ? n : lo);
    stop = hi < 0 ? (hi + n < 0 ? 0 : hi + n) : (hi > n ? n : hi);
}

// does box k = (cx, cy, w, h) normalised to the image cover pixel (x, y) of the H x W map?
// (monodetr.py:451-456 scales by (W, H, W, H) and converts to corners; balancer.py:60-64: floor / ceil)
MDETR_HD bool ddn_covers(const float *box, int x, int y, int H, int W)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float cx = box[0] * static_cast<float>(W), cy = box[1] * static_cast<float>(H);
    const float bw = box[2] * static_cast<float>(W), bh = box[3] * static_cast<float>(H);
    const float hx = 0.5f * bw, hy = 0.5f * bh;
    const int u1 = static_cast<int>(ddn_floor(cx - hx)), v1 = static_cast<int>(ddn_floor(cy - hy));
    const int u2 = static_cast<int>(ddn_ceil(cx + hx)), v2 = static_cast<int>(ddn_ceil(cy + hy));
    int x0, x1, y0, y1;
    ddn_slice(u1, u2, W, x0, x1);
    ddn_slice(v1, v2, H, y0, y1);
    return x >= x0 && x < x1 && y >= y0 && y < y1;
}

// target bin (ddn_loss.py:64-101, mode LID, target=True) and foreground flag of one pixel
MDETR_HD int ddn_target(const DdnDims &d, const float *boxes_b, const float *depth_b, const unsigned char *valid_b,
                        int x, int y, bool &fg)
{
    float nearest = 0.f;
    bool any = false;
    // (batches of 5 boxes, every load of a batch before its first use and no branch between them: one box per iteration behind
    // `if (!valid) continue` was K dependent round trips per pixel)
    for (int k0 = 0; k0 < d.K; k0 += 5) {
        float bx[5][4], dp[5];
        unsigned char vl[5];
        for (int u = 0; u < 5; ++u) {
            const int k = k0 + u < d.K ? k0 + u : d.K - 1;
            vl[u] = valid_b[k];
            dp[u] = depth_b[k];
            for (int i = 0; i < 4; ++i) bx[u][i] = boxes_b[4 * k + i];
        }
        for (int u = 0; u < 5; ++u) {
            const bool hit = (k0 + u < d.K) & (vl[u] != 0) & ddn_covers(bx[u], x, y, d.H, d.W);
            nearest = hit ? (any ? (dp[u] < nearest ? dp[u] : nearest) : dp[u]) : nearest;
            any = any | hit;
        }
    }
    fg = any;
    const int nbins = d.C - 1;
    const float bin_size = 2.f * (d.depth_max - d.depth_min) / (static_cast<float>(nbins) * (1.f + static_cast<float>(nbins)));
    const float idx = -0.5f + 0.5f * ddn_sqrt(1.f + 8.f * ((any ? nearest : 0.f) - d.depth_min) / bin_size);
    if (!(idx >= 0.f) || idx > static_cast<float>(nbins) || idx != idx) return nbins;     // also catches NaN
    return static_cast<int>(idx);                                                        // truncation, as .type(int64)
}

// focal loss of one pixel (focalloss.py:58-129, gamma = 2): value, and (if g != nullptr) d loss / d logit_c
// scaled by `scale`, written with stride sc.
//   focal_c = -alpha (1 - p_c)^2 log p_c ;  loss = focal_t + 1e-6 sum_c focal_c
constexpr int kDdnRegs = 96;              // classes (depth bins + 1) a pixel keeps in registers: 81 in the reference's configuration

template <typename Z>
MDETR_HD float ddn_pixel_body(const DdnDims &d, const Z &zc, int t, float scale, float *g)
{
    float mx = zc(0);
    for (int c = 1; c < d.C; ++c) mx = zc(c) > mx ? zc(c) : mx;
    float se = 0.f;
    for (int c = 0; c < d.C; ++c) se += ddn_exp(zc(c) - mx);
    const float lse = mx + ddn_log(se);
    float loss = 0.f, sum_a = 0.f, a_t = 0.f;
    for (int c = 0; c < d.C; ++c) {
        const float logp = zc(c) - lse, p = ddn_exp(logp), om = 1.f - p;
        const float focal = -d.alpha * om * om * logp;
        loss += (c == t ? 1.f : 0.f) * focal + 1e-6f * focal;
        const float a = d.alpha * (2.f * om * p * logp - om * om);        // d focal_c / d p_c * p_c
        sum_a += a;
        if (c == t) a_t = a;
    }
    if (g) {
        const float tail = a_t + 1e-6f * sum_a;
        for (int c = 0; c < d.C; ++c) {
            const float logp = zc(c) - lse, p = ddn_exp(logp), om = 1.f - p;
            const float a = d.alpha * (2.f * om * p * logp - om * om);
            g[c * d.sc] = scale * ((c == t ? a_t : 0.f) + 1e-6f * a - p * tail);
        }
    }
    return loss;
}

MDETR_HD float ddn_pixel(const DdnDims &d, const float *z, int t, float scale, float *g)
{
    if (d.C <= kDdnRegs) {
        // the pixel's logits ONCE into registers, 16 loads at a time before their first use (a class beyond C re-reads the last one: no
        // branch around a load): the three passes below then run on registers -- from memory they were 3 x 81 loads in a chain of
        // short batches per pixel
        float zz[kDdnRegs];
#pragma unroll
        for (int c0 = 0; c0 < kDdnRegs; c0 += 16) {
            if (c0 >= d.C) break;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int c = c0 + u < d.C ? c0 + u : d.C - 1;
                zz[c0 + u] = z[c * d.sc];
            }
        }
        // (static register indices: the loops over c are unrolled to kDdnRegs and cut at C)
        float mx = zz[0];
#pragma unroll
        for (int c = 1; c < kDdnRegs; ++c) if (c < d.C) mx = zz[c] > mx ? zz[c] : mx;
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < kDdnRegs; ++c) if (c < d.C) se += ddn_exp(zz[c] - mx);
        const float lse = mx + ddn_log(se);
        float loss = 0.f, sum_a = 0.f, a_t = 0.f;
#pragma unroll
        for (int c = 0; c < kDdnRegs; ++c) {
            if (c < d.C) {
                const float logp = zz[c] - lse, p = ddn_exp(logp), om = 1.f - p;
                const float focal = -d.alpha * om * om * logp;
                loss += (c == t ? 1.f : 0.f) * focal + 1e-6f * focal;
                const float a = d.alpha * (2.f * om * p * logp - om * om);
                sum_a += a;
                if (c == t) a_t = a;
            }
        }
        if (g) {
            const float tail = a_t + 1e-6f * sum_a;
#pragma unroll
            for (int c = 0; c < kDdnRegs; ++c) {
                if (c < d.C) {
                    const float logp = zz[c] - lse, p = ddn_exp(logp), om = 1.f - p;
                    const float a = d.alpha * (2.f * om * p * logp - om * om);
                    g[c * d.sc] = scale * ((c == t ? a_t : 0.f) + 1e-6f * a - p * tail);
                }
            }
        }
        return loss;
    }
    return ddn_pixel_body(d, [&](int c) { return z[c * d.sc]; }, t, scale, g);
}

}  // namespace mdetr
