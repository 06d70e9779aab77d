// tests/native/host_kernels.cpp -- HOST build (g++) of the per-element arithmetic the HIP kernels use,
// so that CPU tests can validate it against recorded reference results without a GPU.  Test
// infrastructure only: nothing in the product links or loads this.
#include <stdint.h>
#include <string.h>

#include "../../monodetr_amd/csrc/adamw_math.h"
#include "../../monodetr_amd/csrc/ddn_loss_math.h"
#include "../../monodetr_amd/csrc/kitti_prep_math.h"
#include "../../monodetr_amd/csrc/msda_prologue_math.h"
#include "../../monodetr_amd/csrc/pair_losses_math.h"

namespace {

inline float bf16_to_f32(uint16_t h)
{
    uint32_t u = static_cast<uint32_t>(h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline uint16_t f32_to_bf16(float f)                    // round to nearest even, as __float2bfloat16
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return static_cast<uint16_t>((u >> 16) | 0x40);   // NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return static_cast<uint16_t>(u >> 16);
}

}  // namespace

extern "C" {

// same argument list as mdetr_adamw_step (include/monodetr_amd.h); device / stream are ignored
int mdetr_adamw_step(int param_dtype, void *param, float *master, const void *grad, float *exp_avg, float *exp_avg_sq,
                     int64_t n, int64_t n_no_decay, float beta1, float beta2, float eps, float weight_decay,
                     float step_size, const float *step_size_dev, int device, void *stream)
{
    (void)device; (void)stream;
    const mdetr::AdamWCoef c{beta1, beta2, eps, weight_decay};
    const float step = step_size_dev ? *step_size_dev : step_size;
    for (int64_t i = 0; i < n; ++i) {
        const float g = param_dtype == 2 ? bf16_to_f32(static_cast<const uint16_t *>(grad)[i])
                                         : static_cast<const float *>(grad)[i];
        float m = exp_avg[i], v = exp_avg_sq[i];
        const float q = mdetr::adamw_element(master[i], g, m, v, c, i < n_no_decay ? 0.f : weight_decay, step);
        exp_avg[i] = m; exp_avg_sq[i] = v; master[i] = q;
        if (param_dtype == 2) static_cast<uint16_t *>(param)[i] = f32_to_bf16(q);
    }
    return 0;
}

// mdetr_msda_prologue_forward / _backward: serial loops over the (image, query, head) units
static float host_ld(int dt, const void *p, int64_t i)
{
    return dt == 2 ? bf16_to_f32(static_cast<const uint16_t *>(p)[i]) : static_cast<const float *>(p)[i];
}
static void host_st(int dt, void *p, int64_t i, float v)
{
    if (dt == 2) static_cast<uint16_t *>(p)[i] = f32_to_bf16(v); else static_cast<float *>(p)[i] = v;
}

int mdetr_msda_prologue_forward(int io_dtype, int ref_dtype, const void *offsets, const void *logits, const void *ref,
                                const int64_t *spatial_shapes, float *sampling_loc, float *attn_weight,
                                int B, int Lq, int M, int L, int P, int R, int64_t ref_sb, int64_t ref_sq, int64_t ref_sl,
                                int device, void *stream)
{
    (void)device; (void)stream;
    const int LP = L * P;
    for (int b = 0; b < B; ++b)
        for (int q = 0; q < Lq; ++q)
            for (int m = 0; m < M; ++m) {
                const int64_t u = (static_cast<int64_t>(b) * Lq + q) * M + m;
                float lg[64], at[64];
                for (int i = 0; i < LP; ++i) lg[i] = host_ld(io_dtype, logits, u * LP + i);
                mdetr::pro_softmax(lg, LP, at);
                for (int i = 0; i < LP; ++i) attn_weight[u * LP + i] = at[i];
                for (int l = 0; l < L; ++l) {
                    float rl[6];
                    for (int r = 0; r < R; ++r) rl[r] = host_ld(ref_dtype, ref, b * ref_sb + q * ref_sq + l * ref_sl + r);
                    const float wh[2] = {static_cast<float>(spatial_shapes[2 * l + 1]), static_cast<float>(spatial_shapes[2 * l])};
                    for (int p = 0; p < P; ++p)
                        for (int c = 0; c < 2; ++c) {
                            const int64_t i = (u * LP + l * P + p) * 2 + c;
                            sampling_loc[i] = mdetr::pro_location(host_ld(io_dtype, offsets, i), rl, R, c, wh[c], P);
                        }
                }
            }
    return 0;
}

int mdetr_msda_prologue_backward(int io_dtype, int ref_dtype, const void *offsets, const void *ref, const int64_t *spatial_shapes,
                                 const float *attn_weight, const float *grad_loc, const float *grad_attn,
                                 void *grad_offsets, void *grad_logits, float *grad_ref,
                                 int B, int Lq, int M, int L, int P, int R, int64_t ref_sb, int64_t ref_sq, int64_t ref_sl,
                                 int device, void *stream)
{
    (void)device; (void)stream;
    const int LP = L * P;
    if (grad_ref) memset(grad_ref, 0, sizeof(float) * static_cast<size_t>(B) * Lq * L * R);
    for (int b = 0; b < B; ++b)
        for (int q = 0; q < Lq; ++q)
            for (int m = 0; m < M; ++m) {
                const int64_t u = (static_cast<int64_t>(b) * Lq + q) * M + m;
                float gl[64];
                mdetr::pro_softmax_backward(attn_weight + u * LP, grad_attn + u * LP, LP, gl);
                for (int i = 0; i < LP; ++i) host_st(io_dtype, grad_logits, u * LP + i, gl[i]);
                for (int l = 0; l < L; ++l) {
                    float rl[6], gr[6] = {0, 0, 0, 0, 0, 0};
                    for (int r = 0; r < R; ++r) rl[r] = host_ld(ref_dtype, ref, b * ref_sb + q * ref_sq + l * ref_sl + r);
                    const float wh[2] = {static_cast<float>(spatial_shapes[2 * l + 1]), static_cast<float>(spatial_shapes[2 * l])};
                    for (int p = 0; p < P; ++p)
                        for (int c = 0; c < 2; ++c) {
                            const int64_t i = (u * LP + l * P + p) * 2 + c;
                            const float off = R == 2 ? 0.f : host_ld(io_dtype, offsets, i);
                            host_st(io_dtype, grad_offsets, i, mdetr::pro_location_backward(grad_loc[i], off, rl, R, c, wh[c], P, grad_ref ? gr : nullptr));
                        }
                    if (grad_ref)
                        for (int r = 0; r < R; ++r) grad_ref[((static_cast<int64_t>(b) * Lq + q) * L + l) * R + r] += gr[r];
                }
            }
    return 0;
}

// mdetr_lsa_forward_fused: the kernel's cost arithmetic (pl_match_cost) + a serial version of its
// shortest-augmenting-path solver (float64 on the fp32 costs, smallest column index on ties)
int mdetr_lsa_forward_fused(const float *logits, const float *boxes, const int64_t *labels, const float *boxes3d,
                            const int32_t *num_targets, int32_t *assign, int layers, int images, int groups, int n,
                            int kmax, int num_classes, float w_class, float w_bbox, float w_center, float w_giou,
                            float focal_alpha, int device, void *stream)
{
    (void)device; (void)stream;
    const mdetr::MatchWeights mw{w_class, w_bbox, w_center, w_giou, focal_alpha};
    const double INF = 1e300;
    for (int li = 0; li < layers * images; ++li)
        for (int g = 0; g < groups; ++g) {
            const int image = li % images;
            int k = num_targets[image];
            k = k < 0 ? 0 : (k > kmax ? kmax : k);
            int32_t *out = assign + (static_cast<int64_t>(li) * groups + g) * kmax;
            for (int t = 0; t < kmax; ++t) out[t] = -1;
            if (k == 0) continue;
            double a[64][64], u[64] = {0}, v[64] = {0};
            int p[64];
            for (int j = 0; j < n; ++j) {
                p[j] = -1;
                const int64_t row = (static_cast<int64_t>(li) * groups + g) * n + j;
                for (int t = 0; t < k; ++t) {
                    const int64_t tk = static_cast<int64_t>(image) * kmax + t;
                    a[t][j] = static_cast<double>(mdetr::pl_match_cost(logits + row * num_classes, boxes + row * 6,
                                                                       static_cast<int>(labels[tk]), boxes3d + tk * 6, mw));
                }
            }
            for (int i = 0; i < k; ++i) {
                double minv[64];
                int way[64];
                bool used[64];
                for (int j = 0; j < n; ++j) { minv[j] = INF; way[j] = -2; used[j] = false; }
                int j0 = -1;
                while (true) {
                    if (j0 >= 0) used[j0] = true;
                    const int i0 = j0 < 0 ? i : p[j0];
                    double delta = INF;
                    int j1 = -1;
                    for (int j = 0; j < n; ++j) {
                        if (used[j]) continue;
                        const double cur = a[i0][j] - u[i0] - v[j];
                        if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                        if (minv[j] < delta) { delta = minv[j]; j1 = j; }
                    }
                    for (int j = 0; j < n; ++j) {
                        if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
                        else minv[j] -= delta;
                    }
                    u[i] += delta;
                    j0 = j1;
                    if (p[j0] < 0) break;
                }
                while (j0 >= 0) {
                    const int jp = way[j0];
                    p[j0] = jp < 0 ? i : p[jp];
                    j0 = jp;
                }
            }
            for (int j = 0; j < n; ++j)
                if (p[j] >= 0) out[p[j]] = g * n + j;
        }
    return 0;
}

// same argument lists as mdetr_ddn_loss_forward / _backward, serial loops over the pixels
static mdetr::DdnDims host_ddn_dims(int B, int C, int H, int W, int K, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                                    float alpha, float fg_weight, float bg_weight, float depth_min, float depth_max)
{
    return mdetr::DdnDims{B, C, H, W, K, sb, sc, sh, sw, alpha, fg_weight, bg_weight, depth_min, depth_max};
}

int mdetr_ddn_loss_forward(const float *logits, const float *boxes, const float *depth, const uint8_t *valid,
                           int B, int C, int H, int W, int K, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                           float alpha, float fg_weight, float bg_weight, float depth_min, float depth_max,
                           float *out, void *workspace, int device, void *stream)
{
    (void)workspace; (void)device; (void)stream;
    const mdetr::DdnDims d = host_ddn_dims(B, C, H, W, K, sb, sc, sh, sw, alpha, fg_weight, bg_weight, depth_min, depth_max);
    double sum = 0.0;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                bool fg;
                const int t = mdetr::ddn_target(d, boxes + static_cast<int64_t>(b) * K * 4, depth + static_cast<int64_t>(b) * K,
                                                valid + static_cast<int64_t>(b) * K, x, y, fg);
                sum += mdetr::ddn_pixel(d, logits + b * sb + y * sh + x * sw, t, 0.f, nullptr) * (fg ? fg_weight : bg_weight);
            }
    out[0] = static_cast<float>(sum / (static_cast<double>(B) * H * W));
    return 0;
}

int mdetr_ddn_loss_backward(const float *logits, const float *boxes, const float *depth, const uint8_t *valid,
                            int B, int C, int H, int W, int K, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                            float alpha, float fg_weight, float bg_weight, float depth_min, float depth_max,
                            const float *grad_out, float *grad_logits, int device, void *stream)
{
    (void)device; (void)stream;
    const mdetr::DdnDims d = host_ddn_dims(B, C, H, W, K, sb, sc, sh, sw, alpha, fg_weight, bg_weight, depth_min, depth_max);
    const float n = static_cast<float>(static_cast<int64_t>(B) * H * W);
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                bool fg;
                const int t = mdetr::ddn_target(d, boxes + static_cast<int64_t>(b) * K * 4, depth + static_cast<int64_t>(b) * K,
                                                valid + static_cast<int64_t>(b) * K, x, y, fg);
                const int64_t off = b * sb + y * sh + x * sw;
                mdetr::ddn_pixel(d, logits + off, t, grad_out[0] * (fg ? fg_weight : bg_weight) / n, grad_logits + off);
            }
    return 0;
}

// same argument lists as mdetr_pair_losses_forward / _backward (include/monodetr_amd.h), serial loops over the rows
int64_t mdetr_pair_losses_workspace_bytes(int L, int B) { return 16 + 0 * (L + B); }

int mdetr_pair_losses_forward(const float *logits, const float *boxes, const float *dims, const float *depths,
                              const float *angles, const int32_t *assign, const int64_t *labels, const float *boxes3d,
                              const float *depth, const float *size3d, const int64_t *heading_bin,
                              const float *heading_res, const uint8_t *valid, const int32_t *num,
                              int L, int B, int Q, int C, int G, int K, float focal_alpha,
                              float num_boxes, const float *num_boxes_dev, float *out, float *comp, void *workspace,
                              int device, void *stream)
{
    (void)workspace; (void)device; (void)stream;
    using namespace mdetr;
    const PairLossDims d{L, B, Q, C, G, K, focal_alpha};
    const PairLossIn in{logits, boxes, dims, depths, angles, assign, reinterpret_cast<const long long *>(labels),
                        reinterpret_cast<const long long *>(heading_bin), boxes3d, depth, size3d, heading_res, valid};
    const float nb = num_boxes_dev ? *num_boxes_dev : num_boxes;
    for (int l = 0; l < L; ++l) {
        float acc[kPairLossRows + 3] = {0};
        float card_err = 0.f;
        for (int b = 0; b < B; ++b) {
            int fg = 0;
            for (int q = 0; q < Q; ++q) fg += pl_row_forward(d, in, l, b, q, acc) ? 1 : 0;
            const float diff = static_cast<float>(fg) - static_cast<float>(num[b]);
            card_err += diff < 0.f ? -diff : diff;
        }
        const float s_rel = acc[kPairLossRows], cf = acc[kLossDim] / (s_rel > 1e-12f ? s_rel : 1e-12f);
        comp[l] = cf;
        for (int r = 0; r < kNumWeighted; ++r) out[r * L + l] = acc[r] / nb;
        out[kLossDim * L + l] = s_rel * cf / nb;
        const float hits = acc[kPairLossRows + 1], nmatch = acc[kPairLossRows + 2];
        out[kClassError * L + l] = 100.f - (nmatch > 0.f ? hits * 100.f / nmatch : 0.f);
        out[kCardinality * L + l] = card_err / static_cast<float>(B);
    }
    return 0;
}

int mdetr_pair_losses_backward(const float *logits, const float *boxes, const float *dims, const float *depths,
                               const float *angles, const int32_t *assign, const int64_t *labels, const float *boxes3d,
                               const float *depth, const float *size3d, const int64_t *heading_bin,
                               const float *heading_res, const uint8_t *valid,
                               int L, int B, int Q, int C, int G, int K, float focal_alpha,
                               float num_boxes, const float *num_boxes_dev, const float *grad_out, const float *comp,
                               float *g_logits, float *g_boxes, float *g_dims, float *g_depths, float *g_angles,
                               int device, void *stream)
{
    (void)device; (void)stream;
    using namespace mdetr;
    const PairLossDims d{L, B, Q, C, G, K, focal_alpha};
    const PairLossIn in{logits, boxes, dims, depths, angles, assign, reinterpret_cast<const long long *>(labels),
                        reinterpret_cast<const long long *>(heading_bin), boxes3d, depth, size3d, heading_res, valid};
    const float inv = 1.f / (num_boxes_dev ? *num_boxes_dev : num_boxes);
    for (int l = 0; l < L; ++l) {
        float w[kNumWeighted];
        for (int i = 0; i < kNumWeighted; ++i) w[i] = grad_out[i * L + l] * inv;
        for (int b = 0; b < B; ++b)
            for (int q = 0; q < Q; ++q)
                pl_row_backward(d, in, l, b, q, w, comp[l], g_logits, g_boxes, g_dims, g_depths, g_angles);
    }
    return 0;
}

// same argument list as mdetr_kitti_preprocess; `pixels`, `images` and `out` are host memory here
int mdetr_kitti_preprocess(const uint8_t *pixels, const MdetrKittiImage *images, int n_images, void *out,
                           int out_dtype, int out_h, int out_w, int channels_last, const float *mean, const float *std,
                           int device, void *stream)
{
    (void)device; (void)stream;
    using namespace mdetr;
    const int64_t plane = static_cast<int64_t>(out_h) * out_w;
    for (int n = 0; n < n_images; ++n)
        for (int oy = 0; oy < out_h; ++oy)
            for (int ox = 0; ox < out_w; ++ox) {
                float px[3];
                kp_pixel(images[n], pixels + images[n].pixel_offset, ox, oy, mean, std, px);
                for (int c = 0; c < 3; ++c) {
                    const int64_t at = channels_last ? (static_cast<int64_t>(n) * plane + static_cast<int64_t>(oy) * out_w + ox) * 3 + c
                                                     : (static_cast<int64_t>(n) * 3 + c) * plane + static_cast<int64_t>(oy) * out_w + ox;
                    if (out_dtype == 0) static_cast<float *>(out)[at] = px[c];
                    else static_cast<uint16_t *>(out)[at] = f32_to_bf16(px[c]);
                }
            }
    return 0;
}

}  // extern "C"
