// monodetr_amd/csrc/pair_losses_math.h -- per-query arithmetic of MonoDETR's set criterion
// (lib/models/monodetr/monodetr.py:320-458: focal classification, 3D-centre L1, box L1 + GIoU, Laplacian
// depth, dimension-aware L1, angle bin + residual, class accuracy, cardinality), values AND analytic
// gradients, shared by the HIP kernels (pair_losses.hip) and by the host build the CPU tests compile with
// g++ (tests/native/host_kernels.cpp) and compare against the PyTorch criterion and its autograd.
//
// Work unit = one query row (decoder level l, image b, query q).  A query is matched to at most one
// ground-truth slot (the Hungarian assignment is a matching within its group), found by scanning the
// K <= 64 slots of its group in `assign[l, b, g, :]`.
#pragma once

#if defined(__HIPCC__)
#define MDETR_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define MDETR_HD inline
#endif

namespace mdetr {

enum PairLossRow {                 // rows of the [kPairLossRows, L] result
    kLossCe = 0, kLossCenter, kLossBbox, kLossGiou, kLossDepth, kLossDim, kLossAngle,
    kClassError, kCardinality, kPairLossRows
};
constexpr int kNumWeighted = 7;    // rows that carry gradients
constexpr int kMaxClasses = 8;
constexpr int kAngleBins = 12;

MDETR_HD float pl_abs(float x) { return x < 0.f ? -x : x; }
MDETR_HD float pl_sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
MDETR_HD float pl_min(float a, float b) { return a < b ? a : b; }
MDETR_HD float pl_max(float a, float b) { return a > b ? a : b; }
MDETR_HD float pl_exp(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __expf(x);
#else
    return std::exp(x);
#endif
}
MDETR_HD float pl_log(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __logf(x);
#else
    return std::log(x);
#endif
}
MDETR_HD float pl_log1p(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return log1pf(x);
#else
    return std::log1p(x);
#endif
}

// slot matched to query q: k in [0, K) or -1.  `assign_g` = assign[l, b, q / n, :], `valid_b` = valid[b, :]
MDETR_HD int pl_find_match(const int *assign_g, const unsigned char *valid_b, int q, int K)
{
    // (batches of 8 slots, every load of a batch before its first use and no branch between them: one slot per iteration with a
    // short-circuit `&&` was 50 dependent round trips ahead of everything else a row does)
    int hit = -1;
    for (int k0 = 0; k0 < K; k0 += 8) {
        int a[8];
        unsigned char v[8];
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + u < K ? k0 + u : K - 1;
            a[u] = assign_g[k];
            v[u] = valid_b[k];
        }
        for (int u = 0; u < 8; ++u)
            hit = (k0 + u < K) & (a[u] == q) & (v[u] != 0) ? k0 + u : hit;
    }
    return hit;
}

// sigmoid focal loss on one logit (lib/losses/focal_loss.py:69-94, gamma = 2): value and d/dx
//   ce = BCE-with-logits(x, t);  loss = a_t * ce * (1 - p_t)^2
MDETR_HD void pl_focal(float x, bool t, float alpha, float &loss, float &dx)
{
    const float p = 1.f / (1.f + pl_exp(-x));
    const float ce = pl_max(x, 0.f) - (t ? x : 0.f) + pl_log1p(pl_exp(-pl_abs(x)));
    if (t) {
        const float om = 1.f - p;                                   // 1 - p_t
        const float a = alpha >= 0.f ? alpha : 1.f;
        loss = a * ce * om * om;
        dx = -a * om * om * (2.f * p * ce + om);
    } else {
        const float a = alpha >= 0.f ? 1.f - alpha : 1.f;
        loss = a * ce * p * p;
        dx = a * p * p * (2.f * (1.f - p) * ce + p);
    }
}

// (cx, cy, l, r, t, b) -> (x0, y0, x1, y1)   (utils/box_ops.py:20-24)
MDETR_HD void pl_xyxy(const float *c, float *o)
{
    o[0] = c[0] - c[2]; o[1] = c[1] - c[4]; o[2] = c[0] + c[3]; o[3] = c[1] + c[5];
}

// 1 - GIoU(src, tgt) of one matched pair and its gradient w.r.t. the six source parameters
// (monodetr.py:376-386, utils/box_ops.py:51-72).  Gradient conventions as torch's: minimum / maximum pass
// the gradient to the selected operand (ties are measure-zero for real pairs), clamp(min=0) passes it
// where its input is >= 0.
MDETR_HD void pl_giou(const float *src6, const float *tgt6, float &loss, float *g6)
{
    float s[4], t[4];
    pl_xyxy(src6, s);
    pl_xyxy(tgt6, t);
    const float iw_raw = pl_min(s[2], t[2]) - pl_max(s[0], t[0]), ih_raw = pl_min(s[3], t[3]) - pl_max(s[1], t[1]);
    const float iw = pl_max(iw_raw, 0.f), ih = pl_max(ih_raw, 0.f);
    const float inter = iw * ih;
    const float sw = s[2] - s[0], sh = s[3] - s[1];
    const float area_s = sw * sh, area_t = (t[2] - t[0]) * (t[3] - t[1]);
    const float uni = area_s + area_t - inter;
    const float hw_raw = pl_max(s[2], t[2]) - pl_min(s[0], t[0]), hh_raw = pl_max(s[3], t[3]) - pl_min(s[1], t[1]);
    const float hw = pl_max(hw_raw, 0.f), hh = pl_max(hh_raw, 0.f);
    const float hull = hw * hh;
    const float giou = inter / uni - (hull - uni) / hull;
    loss = 1.f - giou;
    // d giou / d(inter, uni, hull):  giou = inter/uni - 1 + uni/hull
    const float d_inter = 1.f / uni, d_uni = -inter / (uni * uni) + 1.f / hull, d_hull = -uni / (hull * hull);
    // chain to (x0, y0, x1, y1) of the source box
    float gx[4] = {0.f, 0.f, 0.f, 0.f};
    const float d_iw = (d_inter - d_uni) * ih * (iw_raw >= 0.f ? 1.f : 0.f);     // inter enters uni with -1
    const float d_ih = (d_inter - d_uni) * iw * (ih_raw >= 0.f ? 1.f : 0.f);
    if (s[2] <= t[2]) gx[2] += d_iw;                                                // min(sx1, tx1)
    if (s[0] >= t[0]) gx[0] -= d_iw;                                                // -max(sx0, tx0)
    if (s[3] <= t[3]) gx[3] += d_ih;
    if (s[1] >= t[1]) gx[1] -= d_ih;
    gx[0] += d_uni * (-sh); gx[2] += d_uni * sh;                                    // area_s
    gx[1] += d_uni * (-sw); gx[3] += d_uni * sw;
    const float d_hw = d_hull * hh * (hw_raw >= 0.f ? 1.f : 0.f), d_hh = d_hull * hw * (hh_raw >= 0.f ? 1.f : 0.f);
    if (s[2] >= t[2]) gx[2] += d_hw;                                                // max(sx1, tx1)
    if (s[0] <= t[0]) gx[0] -= d_hw;                                                // -min(sx0, tx0)
    if (s[3] >= t[3]) gx[3] += d_hh;
    if (s[1] <= t[1]) gx[1] -= d_hh;
    // loss = 1 - giou;  x0 = cx - l, y0 = cy - t, x1 = cx + r, y1 = cy + b
    g6[0] = -(gx[0] + gx[2]);
    g6[1] = -(gx[1] + gx[3]);
    g6[2] = gx[0];
    g6[3] = -gx[2];
    g6[4] = gx[1];
    g6[5] = -gx[3];
}

// GIoU value only (the matcher's cost term)
MDETR_HD float pl_giou_value(const float *src6, const float *tgt6)
{
    float s[4], t[4];
    pl_xyxy(src6, s);
    pl_xyxy(tgt6, t);
    const float iw = pl_max(pl_min(s[2], t[2]) - pl_max(s[0], t[0]), 0.f), ih = pl_max(pl_min(s[3], t[3]) - pl_max(s[1], t[1]), 0.f);
    const float inter = iw * ih;
    const float uni = (s[2] - s[0]) * (s[3] - s[1]) + (t[2] - t[0]) * (t[3] - t[1]) - inter;
    const float hw = pl_max(pl_max(s[2], t[2]) - pl_min(s[0], t[0]), 0.f), hh = pl_max(pl_max(s[3], t[3]) - pl_min(s[1], t[1]), 0.f);
    const float hull = hw * hh;
    return inter / uni - (hull - uni) / hull;
}

struct MatchWeights { float w_class, w_bbox, w_center, w_giou, alpha; };

// Hungarian matching cost of (query, target) -- lib/models/monodetr/matcher.py:55-84:
//   w_bbox * L1(l,r,t,b) + w_center * L1(cx,cy) + w_class * (pos - neg focal cost at the target label) - w_giou * GIoU
MDETR_HD float pl_match_cost(const float *logits, const float *box6, int label, const float *tgt6, const MatchWeights &w)
{
    const float p = 1.f / (1.f + pl_exp(-logits[label]));
    const float neg = (1.f - w.alpha) * (p * p) * (-pl_log(1.f - p + 1e-8f));
    const float pos = w.alpha * ((1.f - p) * (1.f - p)) * (-pl_log(p + 1e-8f));
    const float c_center = pl_abs(box6[0] - tgt6[0]) + pl_abs(box6[1] - tgt6[1]);
    const float c_bbox = pl_abs(box6[2] - tgt6[2]) + pl_abs(box6[3] - tgt6[3]) + pl_abs(box6[4] - tgt6[4]) + pl_abs(box6[5] - tgt6[5]);
    return w.w_bbox * c_bbox + w.w_center * c_center + w.w_class * (pos - neg) + w.w_giou * (-pl_giou_value(box6, tgt6));
}

struct PairLossDims {
    int L, B, Q, C, G, K;          // levels, images, queries, classes, groups, target slots
    float alpha;                   // focal alpha
};

struct PairLossIn {                // level-stacked predictions [L, B, Q, .] fp32 and padded ground truth [B, K, .]
    const float *logits, *boxes, *dims, *depths, *angles;     // [..,C] [..,6] [..,3] [..,2] [..,24]
    const int *assign;                                        // [L, B, G, K] matched query or -1
    const long long *labels, *heading_bin;                    // [B, K] int64
    const float *boxes3d, *depth, *size3d, *heading_res;      // [B,K,6] [B,K] [B,K,3] [B,K]
    const unsigned char *valid;                               // [B, K]
};

// forward contributions of one row, added into acc[kPairLossRows + 3]:
//   acc[row] += un-normalised loss;  acc[kLossDim] = sum |d|;  extra slots: [kPairLossRows] = sum relative |d|,
//   [kPairLossRows + 1] = correctly classified matched queries, [kPairLossRows + 2] = matched queries
// returns whether argmax(logits) is a foreground class (for the cardinality count)
MDETR_HD bool pl_row_forward(const PairLossDims &d, const PairLossIn &in, int l, int b, int q, float *acc)
{
    const int n = d.Q / d.G;
    const long long row = (static_cast<long long>(l) * d.B + b) * d.Q + q;
    const int *ag = in.assign + ((static_cast<long long>(l) * d.B + b) * d.G + q / n) * d.K;
    const unsigned char *vb = in.valid + static_cast<long long>(b) * d.K;
    const int k = pl_find_match(ag, vb, q, d.K);
    const long long tk = static_cast<long long>(b) * d.K + (k < 0 ? 0 : k);
    const int label = k < 0 ? -1 : static_cast<int>(in.labels[tk]);
    const float *x = in.logits + row * d.C;
    int best = 0;
    for (int c = 0; c < d.C; ++c) {
        float lo, dx;
        pl_focal(x[c], c == label, d.alpha, lo, dx);
        acc[kLossCe] += lo;
        if (x[c] > x[best]) best = c;
    }
    if (k >= 0) {
        const float *bx = in.boxes + row * 6, *tb = in.boxes3d + tk * 6;
        acc[kLossCenter] += pl_abs(bx[0] - tb[0]) + pl_abs(bx[1] - tb[1]);
        acc[kLossBbox] += pl_abs(bx[2] - tb[2]) + pl_abs(bx[3] - tb[3]) + pl_abs(bx[4] - tb[4]) + pl_abs(bx[5] - tb[5]);
        float gl, g6[6];
        pl_giou(bx, tb, gl, g6);
        acc[kLossGiou] += gl;
        const float *dp = in.depths + row * 2;
        acc[kLossDepth] += 1.4142f * pl_exp(-dp[1]) * pl_abs(dp[0] - in.depth[tk]) + dp[1];
        const float *dm = in.dims + row * 3, *ts = in.size3d + tk * 3;
        for (int i = 0; i < 3; ++i) {
            const float df = pl_abs(dm[i] - ts[i]);
            acc[kLossDim] += df;
            acc[kPairLossRows] += df / ts[i];
        }
        const float *an = in.angles + row * 24;
        const int bin = static_cast<int>(in.heading_bin[tk]);
        float mx = an[0];
        for (int i = 1; i < kAngleBins; ++i) mx = pl_max(mx, an[i]);
        float se = 0.f;
        for (int i = 0; i < kAngleBins; ++i) se += pl_exp(an[i] - mx);
        acc[kLossAngle] += (mx + pl_log(se) - an[bin]) + pl_abs(an[kAngleBins + bin] - in.heading_res[tk]);
        acc[kPairLossRows + 1] += best == label ? 1.f : 0.f;
        acc[kPairLossRows + 2] += 1.f;
    }
    return best != d.C - 1;
}

// gradients of one row.  w[r] = upstream gradient of result row r at level l, already divided by
// num_boxes; comp = sum|d| / sum relative|d| of level l (the dimension-aware L1's detached factor).
MDETR_HD void pl_row_backward(const PairLossDims &d, const PairLossIn &in, int l, int b, int q, const float *w, float comp,
                              float *g_logits, float *g_boxes, float *g_dims, float *g_depths, float *g_angles)
{
    const int n = d.Q / d.G;
    const long long row = (static_cast<long long>(l) * d.B + b) * d.Q + q;
    const int *ag = in.assign + ((static_cast<long long>(l) * d.B + b) * d.G + q / n) * d.K;
    const unsigned char *vb = in.valid + static_cast<long long>(b) * d.K;
    const int k = pl_find_match(ag, vb, q, d.K);
    const long long tk = static_cast<long long>(b) * d.K + (k < 0 ? 0 : k);
    const int label = k < 0 ? -1 : static_cast<int>(in.labels[tk]);
    const float *x = in.logits + row * d.C;
    for (int c = 0; c < d.C; ++c) {
        float lo, dx;
        pl_focal(x[c], c == label, d.alpha, lo, dx);
        g_logits[row * d.C + c] = w[kLossCe] * dx;
    }
    float *gb = g_boxes + row * 6, *gd = g_dims + row * 3, *gp = g_depths + row * 2, *ga = g_angles + row * 24;
    if (k < 0) {
        for (int i = 0; i < 6; ++i) gb[i] = 0.f;
        for (int i = 0; i < 3; ++i) gd[i] = 0.f;
        gp[0] = gp[1] = 0.f;
        for (int i = 0; i < 24; ++i) ga[i] = 0.f;
        return;
    }
    const float *bx = in.boxes + row * 6, *tb = in.boxes3d + tk * 6;
    float gl, g6[6];
    pl_giou(bx, tb, gl, g6);
    for (int i = 0; i < 6; ++i)
        gb[i] = (i < 2 ? w[kLossCenter] : w[kLossBbox]) * pl_sign(bx[i] - tb[i]) + w[kLossGiou] * g6[i];
    const float *dp = in.depths + row * 2;
    const float e = 1.4142f * pl_exp(-dp[1]), df = dp[0] - in.depth[tk];
    gp[0] = w[kLossDepth] * e * pl_sign(df);
    gp[1] = w[kLossDepth] * (1.f - e * pl_abs(df));
    const float *dm = in.dims + row * 3, *ts = in.size3d + tk * 3;
    for (int i = 0; i < 3; ++i) gd[i] = w[kLossDim] * comp * pl_sign(dm[i] - ts[i]) / ts[i];
    const float *an = in.angles + row * 24;
    const int bin = static_cast<int>(in.heading_bin[tk]);
    float mx = an[0];
    for (int i = 1; i < kAngleBins; ++i) mx = pl_max(mx, an[i]);
    float se = 0.f;
    for (int i = 0; i < kAngleBins; ++i) se += pl_exp(an[i] - mx);
    for (int i = 0; i < kAngleBins; ++i) {
        ga[i] = w[kLossAngle] * (pl_exp(an[i] - mx) / se - (i == bin ? 1.f : 0.f));
        ga[kAngleBins + i] = i == bin ? w[kLossAngle] * pl_sign(an[kAngleBins + bin] - in.heading_res[tk]) : 0.f;
    }
}

}  // namespace mdetr
