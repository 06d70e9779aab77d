// monodetr_amd/csrc/capi.hip -- extern "C" entry points of libmonodetr_amd.so (include/monodetr_amd.h).
// Argument validation + device/stream plumbing + error reporting; kernels live in msda.hip.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <atomic>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/monodetr_amd.h"
#include "attn.h"
#include "adamw.h"
#include "add_ln.h"
#include "bias_act.h"
#include "colsum.h"
#include "group_norm.h"
#include "small_wgrad.h"
#include "decimate.h"
#include "wfold.h"
#include "conv3x3.h"
#include "conv_stem.h"
#include "conv_taps.h"
#include "conv_wgrad.h"
#include "ddn_loss.h"
#include "kitti_prep.h"
#include "lsa.h"
#include "pair_losses.h"
#include "rotate_iou.h"
#include "tgemm.h"
#include "sgemm.h"
#include "head_tail.h"
#include "mdetr_tune.h"
#include "twgrad.h"
#include "msda.h"
#include "msda_prologue.h"
#include "msda_prologue_math.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// Make `device` current for the duration of one call and restore the caller's device afterwards
// (torch tracks the current device per thread; do not leave it changed).
struct DeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device); else prev = -1;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_common(const char *who, int dtype, int B, int S, int M, int D, int L, int Lq, int P)
{
    if (dtype != MDETR_F32 && dtype != MDETR_F64)
        return fail(MDETR_E_ARG, "%s: unsupported dtype %d (MDETR_F32 / MDETR_F64 only, as AT_DISPATCH_FLOATING_TYPES)", who, dtype);
    if (B < 0 || S < 0 || M <= 0 || D <= 0 || L <= 0 || Lq < 0 || P <= 0)
        return fail(MDETR_E_ARG, "%s: bad sizes B=%d S=%d M=%d D=%d L=%d Lq=%d P=%d", who, B, S, M, D, L, Lq, P);
    if (static_cast<int64_t>(S) * M * D >= (1ll << 31) || static_cast<int64_t>(Lq) * M * L * P * 2 >= (1ll << 31))
        return fail(MDETR_E_ARG, "%s: one image exceeds 2^31 elements (S*M*D or Lq*M*L*P*2)", who);
    return MDETR_OK;
}

}  // namespace

// ---- optional kernel timing ------------------------------------------------------------------
namespace mdetr {
namespace {
struct Rec { int kind, Lq; hipEvent_t a, b; double mflop, kbytes; };
std::atomic<int> g_prof_on{0};
std::mutex g_prof_mu;
std::vector<Rec> g_recs;
thread_local Rec *t_open = nullptr;
thread_local Rec t_cur;
}  // namespace

void profile_begin(int kind, int Lq, hipStream_t st)
{
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    t_cur = Rec{kind, Lq, nullptr, nullptr, 0.0, 0.0};
    if (hipEventCreate(&t_cur.a) != hipSuccess || hipEventCreate(&t_cur.b) != hipSuccess) return;
    (void)hipEventRecord(t_cur.a, st);
    t_open = &t_cur;
}

void profile_work(double mflop, double kbytes)
{
    if (!t_open) return;
    t_cur.mflop += mflop;
    t_cur.kbytes += kbytes;
}

void profile_end(hipStream_t st)
{
    if (!t_open) return;
    (void)hipEventRecord(t_cur.b, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_recs.push_back(t_cur);
    t_open = nullptr;
}
}  // namespace mdetr

extern "C" {

int mdetr_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(mdetr::g_prof_mu);
    if (on) {
        for (auto &r : mdetr::g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        mdetr::g_recs.clear();
    }
    mdetr::g_prof_on.store(on ? 1 : 0);
    return MDETR_OK;
}

static int profile_read_impl(double *rows, int cap, int width)
{
    if (!rows || cap < 0) return fail(MDETR_E_ARG, "mdetr_profile_read: bad arguments");
    std::lock_guard<std::mutex> lk(mdetr::g_prof_mu);
    struct Agg { double launches = 0, ms = 0, mflop = 0, kbytes = 0; };
    std::map<std::pair<int, int>, Agg> agg;                           // (kind, key)
    for (auto &r : mdetr::g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess)
            return fail(MDETR_E_HIP, "mdetr_profile_read: event query failed");
        auto &e = agg[{r.kind, r.Lq}];
        e.launches += 1.0;
        e.ms += ms;
        e.mflop += r.mflop;
        e.kbytes += r.kbytes;
    }
    int n = 0;
    for (auto &kv : agg) {
        if (n >= cap) break;
        double *o = rows + static_cast<size_t>(width) * n;
        o[0] = kv.first.first; o[1] = kv.first.second; o[2] = kv.second.launches; o[3] = kv.second.ms;
        if (width == 6) { o[4] = kv.second.mflop; o[5] = kv.second.kbytes; }
        ++n;
    }
    return n;
}

int mdetr_profile_read(double *rows, int cap) { return profile_read_impl(rows, cap, 4); }

int mdetr_profile_read_work(double *rows, int cap) { return profile_read_impl(rows, cap, 6); }

int mdetr_abi_version(void) { return MDETR_ABI_VERSION; }

const char *mdetr_last_error(void) { return g_err; }

int mdetr_msda_variant(int dtype, int M, int D, int L, int P)
{
    (void)M;
    return mdetr::msda_fast_path(dtype, D, L, P) ? 1 : 0;
}

int mdetr_msda_forward(int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                       const void *loc, const void *attn, void *out,
                       int B, int S, int M, int D, int L, int Lq, int P, int device, void *stream)
{
    if (int rc = check_common("mdetr_msda_forward", dtype, B, S, M, D, L, Lq, P)) return rc;
    if (B == 0 || Lq == 0) return MDETR_OK;
    if (!value || !spatial_shapes || !level_start || !loc || !attn || !out)
        return fail(MDETR_E_ARG, "mdetr_msda_forward: null pointer");
    if (!aligned16(value) || !aligned16(loc) || !aligned16(attn) || !aligned16(out))
        return fail(MDETR_E_ALIGN, "mdetr_msda_forward: value/loc/attn/out must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_forward: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::msda_forward_launch(dtype, value, spatial_shapes, level_start, loc, attn, out,
                                                    B, S, M, D, L, Lq, P, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_forward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_msda_backward(int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                        const void *loc, const void *attn, const void *grad_out,
                        void *grad_value, void *grad_loc, void *grad_attn,
                        int B, int S, int M, int D, int L, int Lq, int P, int device, void *stream)
{
    if (int rc = check_common("mdetr_msda_backward", dtype, B, S, M, D, L, Lq, P)) return rc;
    if (B == 0) return MDETR_OK;
    if (!value || !spatial_shapes || !level_start || !grad_value || (Lq && (!loc || !attn || !grad_out || !grad_loc || !grad_attn)))
        return fail(MDETR_E_ARG, "mdetr_msda_backward: null pointer");
    if (!aligned16(value) || !aligned16(loc) || !aligned16(attn) || !aligned16(grad_out) ||
        !aligned16(grad_value) || !aligned16(grad_loc) || !aligned16(grad_attn))
        return fail(MDETR_E_ALIGN, "mdetr_msda_backward: tensors must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_backward: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::msda_backward_launch(dtype, value, spatial_shapes, level_start, loc, attn, grad_out,
                                                     grad_value, grad_loc, grad_attn, B, S, M, D, L, Lq, P,
                                                     static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_backward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_msda_forward_cpu(int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                           const void *loc, const void *attn, void *out, int B, int S, int M, int D, int L, int Lq, int P)
{
    if (int rc = check_common("mdetr_msda_forward_cpu", dtype, B, S, M, D, L, Lq, P)) return rc;
    if (B == 0 || Lq == 0) return MDETR_OK;
    if (!value || !spatial_shapes || !level_start || !loc || !attn || !out)
        return fail(MDETR_E_ARG, "mdetr_msda_forward_cpu: null pointer");
    mdetr::msda_forward_cpu(dtype, value, spatial_shapes, level_start, loc, attn, out, B, S, M, D, L, Lq, P);
    return MDETR_OK;
}

int mdetr_msda_backward_cpu(int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                            const void *loc, const void *attn, const void *grad_out,
                            void *grad_value, void *grad_loc, void *grad_attn, int B, int S, int M, int D, int L, int Lq, int P)
{
    if (int rc = check_common("mdetr_msda_backward_cpu", dtype, B, S, M, D, L, Lq, P)) return rc;
    if (B == 0) return MDETR_OK;
    if (!value || !spatial_shapes || !level_start || !grad_value || (Lq && (!loc || !attn || !grad_out || !grad_loc || !grad_attn)))
        return fail(MDETR_E_ARG, "mdetr_msda_backward_cpu: null pointer");
    mdetr::msda_backward_cpu(dtype, value, spatial_shapes, level_start, loc, attn, grad_out, grad_value, grad_loc, grad_attn,
                             B, S, M, D, L, Lq, P);
    return MDETR_OK;
}

int64_t mdetr_msda_backward_workspace_bytes(int dtype, const int64_t *spatial_shapes_host, const int64_t *level_start_host,
                                            int B, int S, int M, int D, int L, int Lq, int P)
{
    if (dtype != MDETR_F32 || !spatial_shapes_host || !level_start_host) return 0;
    const int64_t a = mdetr::msda_tiled_workspace_bytes(spatial_shapes_host, level_start_host, B, S, M, D, L, Lq, P);
    const int64_t b = mdetr::msda_fused_workspace_bytes(spatial_shapes_host, level_start_host, B, S, M, D, L, Lq, P);
    return a > b ? a : b;
}

int mdetr_msda_backward_ex(int dtype, const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                           const void *loc, const void *attn, const void *grad_out,
                           void *grad_value, void *grad_loc, void *grad_attn,
                           int B, int S, int M, int D, int L, int Lq, int P,
                           const int64_t *spatial_shapes_host, const int64_t *level_start_host,
                           void *workspace, int64_t workspace_bytes, int device, void *stream)
{
    if (int rc = check_common("mdetr_msda_backward_ex", dtype, B, S, M, D, L, Lq, P)) return rc;
    if (B == 0) return MDETR_OK;
    if (!value || !spatial_shapes || !level_start || !grad_value || (Lq && (!loc || !attn || !grad_out || !grad_loc || !grad_attn)))
        return fail(MDETR_E_ARG, "mdetr_msda_backward_ex: null pointer");
    if (!aligned16(value) || !aligned16(loc) || !aligned16(attn) || !aligned16(grad_out) ||
        !aligned16(grad_value) || !aligned16(grad_loc) || !aligned16(grad_attn) || !aligned16(workspace))
        return fail(MDETR_E_ALIGN, "mdetr_msda_backward_ex: tensors must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_backward_ex: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::msda_backward_launch_ex(dtype, value, spatial_shapes, level_start, loc, attn, grad_out,
                                                        grad_value, grad_loc, grad_attn, B, S, M, D, L, Lq, P,
                                                        spatial_shapes_host, level_start_host, workspace, workspace_bytes,
                                                        static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_backward_ex: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

static int attn_check(const char *who, int dtype, const void *q, const void *k, const void *v, int B, int H, int Lq, int Lk,
                      int q_rs, int k_rs, int v_rs)
{
    if (dtype != MDETR_F32 && dtype != MDETR_BF16) return fail(MDETR_E_ARG, "%s: dtype must be MDETR_F32 or MDETR_BF16", who);
    if (B < 0 || H <= 0 || Lq < 0 || Lk < 0) return fail(MDETR_E_ARG, "%s: bad sizes B=%d H=%d Lq=%d Lk=%d", who, B, H, Lq, Lk);
    if (B > 65535 || H > 65535) return fail(MDETR_E_ARG, "%s: B and H must be <= 65535 (grid limits)", who);
    if (B && Lq && Lk && (!q || !k || !v)) return fail(MDETR_E_ARG, "%s: null pointer", who);
    const int align = dtype == MDETR_F32 ? 4 : 8;      // 16-byte vector loads
    if (q_rs % align || k_rs % align || v_rs % align) return fail(MDETR_E_ALIGN, "%s: row strides must keep 16-byte alignment", who);
    if (!aligned16(q) || !aligned16(k) || !aligned16(v)) return fail(MDETR_E_ALIGN, "%s: q/k/v must be 16-byte aligned", who);
    return MDETR_OK;
}

int mdetr_attn_forward(int dtype, const void *q, const void *k, const void *v, const uint8_t *key_padding_mask,
                       void *out, float *lse, int B, int H, int Lq, int Lk,
                       int64_t q_bs, int64_t k_bs, int64_t v_bs, int q_rs, int k_rs, int v_rs,
                       float scale, float dropout_p, uint64_t seed, const uint64_t *seed_dev, int device, void *stream)
{
    if (int rc = attn_check("mdetr_attn_forward", dtype, q, k, v, B, H, Lq, Lk, q_rs, k_rs, v_rs)) return rc;
    if (B == 0 || Lq == 0) return MDETR_OK;
    if (!out || !lse || !aligned16(out)) return fail(MDETR_E_ARG, "mdetr_attn_forward: out / lse null or misaligned");
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail(MDETR_E_ARG, "mdetr_attn_forward: dropout_p must be in [0, 1)");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_attn_forward: set device %d: %s", device, hipGetErrorString(dev.err));
    mdetr::AttnProblem p{dtype, q, k, v, key_padding_mask, B, H, Lq, Lk, q_bs, k_bs, v_bs, q_rs, k_rs, v_rs, scale, dropout_p, seed, seed_dev};
    const hipError_t e = mdetr::attn_forward_launch(p, out, lse, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_attn_forward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_attn_backward(int dtype, const void *q, const void *k, const void *v, const uint8_t *key_padding_mask,
                        const void *out, const void *d_out, const float *lse, float *dsum,
                        void *dq, void *dk, void *dv, int B, int H, int Lq, int Lk,
                        int64_t q_bs, int64_t k_bs, int64_t v_bs, int q_rs, int k_rs, int v_rs,
                        float scale, float dropout_p, uint64_t seed, const uint64_t *seed_dev, int device, void *stream)
{
    if (int rc = attn_check("mdetr_attn_backward", dtype, q, k, v, B, H, Lq, Lk, q_rs, k_rs, v_rs)) return rc;
    if (B == 0) return MDETR_OK;
    if ((Lq && (!out || !d_out || !lse || !dsum || !dq)) || (Lk && (!dk || !dv)))
        return fail(MDETR_E_ARG, "mdetr_attn_backward: null pointer");
    if (!aligned16(out) || !aligned16(d_out) || !aligned16(dq) || !aligned16(dk) || !aligned16(dv))
        return fail(MDETR_E_ALIGN, "mdetr_attn_backward: tensors must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_attn_backward: set device %d: %s", device, hipGetErrorString(dev.err));
    mdetr::AttnProblem p{dtype, q, k, v, key_padding_mask, B, H, Lq, Lk, q_bs, k_bs, v_bs, q_rs, k_rs, v_rs, scale, dropout_p, seed, seed_dev};
    const hipError_t e = mdetr::attn_backward_launch(p, out, d_out, lse, dsum, dq, dk, dv, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_attn_backward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_lsa_forward(const float *cost, const int32_t *num_targets, int32_t *assign,
                      int layers, int images, int groups, int n, int kmax,
                      int64_t img_stride, int64_t q_stride, int64_t t_stride, int device, void *stream)
{
    if (layers < 0 || images < 0 || groups < 0 || n <= 0 || n > 128 || kmax < 0 || kmax > n || kmax > 64)
        return fail(MDETR_E_ARG, "mdetr_lsa_forward: need 0 < n <= 128 and 0 <= kmax <= min(n, 64) (n=%d kmax=%d)", n, kmax);
    if (layers == 0 || images == 0 || groups == 0 || kmax == 0) return MDETR_OK;
    if (!cost || !num_targets || !assign) return fail(MDETR_E_ARG, "mdetr_lsa_forward: null pointer");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_lsa_forward: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::lsa_launch(cost, num_targets, assign, layers, images, groups, n, kmax,
                                           img_stride, q_stride, t_stride, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_lsa_forward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

static int ddn_args(const char *who, int B, int C, int H, int W, int K)
{
    if (B <= 0 || C < 2 || H <= 0 || W <= 0 || K < 0)
        return fail(MDETR_E_ARG, "%s: bad shape B=%d C=%d H=%d W=%d K=%d", who, B, C, H, W, K);
    return MDETR_OK;
}

static mdetr::DdnDims ddn_dims(int B, int C, int H, int W, int K, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                               float alpha, float fg_weight, float bg_weight, float depth_min, float depth_max)
{
    return mdetr::DdnDims{B, C, H, W, K, sb, sc, sh, sw, alpha, fg_weight, bg_weight, depth_min, depth_max};
}

int mdetr_ddn_loss_forward(const float *logits, const float *boxes, const float *depth, const uint8_t *valid,
                           int B, int C, int H, int W, int K, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                           float alpha, float fg_weight, float bg_weight, float depth_min, float depth_max,
                           float *out, void *workspace, int device, void *stream)
{
    if (int rc = ddn_args("mdetr_ddn_loss_forward", B, C, H, W, K)) return rc;
    if (!logits || !out || !workspace || (K > 0 && (!boxes || !depth || !valid)))
        return fail(MDETR_E_ARG, "mdetr_ddn_loss_forward: null pointer");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_ddn_loss_forward: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::ddn_loss_forward_launch(
        ddn_dims(B, C, H, W, K, sb, sc, sh, sw, alpha, fg_weight, bg_weight, depth_min, depth_max), logits, boxes, depth,
        valid, out, workspace, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_ddn_loss_forward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_ddn_loss_backward(const float *logits, const float *boxes, const float *depth, const uint8_t *valid,
                            int B, int C, int H, int W, int K, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                            float alpha, float fg_weight, float bg_weight, float depth_min, float depth_max,
                            const float *grad_out, float *grad_logits, int device, void *stream)
{
    if (int rc = ddn_args("mdetr_ddn_loss_backward", B, C, H, W, K)) return rc;
    if (!logits || !grad_out || !grad_logits || (K > 0 && (!boxes || !depth || !valid)))
        return fail(MDETR_E_ARG, "mdetr_ddn_loss_backward: null pointer");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_ddn_loss_backward: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::ddn_loss_backward_launch(
        ddn_dims(B, C, H, W, K, sb, sc, sh, sw, alpha, fg_weight, bg_weight, depth_min, depth_max), logits, boxes, depth,
        valid, grad_out, grad_logits, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_ddn_loss_backward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

static int pair_losses_args(const char *who, int L, int B, int Q, int C, int G, int K)
{
    if (L <= 0 || B <= 0 || Q <= 0 || C <= 0 || C > mdetr::kMaxClasses || G <= 0 || Q % G != 0 || K <= 0 || K > 64)
        return fail(MDETR_E_ARG, "%s: bad shape L=%d B=%d Q=%d C=%d G=%d K=%d", who, L, B, Q, C, G, K);
    return MDETR_OK;
}

int64_t mdetr_pair_losses_workspace_bytes(int L, int B)
{
    return L <= 0 || B <= 0 ? -1 : mdetr::pair_losses_workspace_bytes(L, B);
}

int mdetr_pair_losses_forward(const float *logits, const float *boxes, const float *dims, const float *depths,
                              const float *angles, const int32_t *assign, const int64_t *labels, const float *boxes3d,
                              const float *depth, const float *size3d, const int64_t *heading_bin,
                              const float *heading_res, const uint8_t *valid, const int32_t *num,
                              int L, int B, int Q, int C, int G, int K, float focal_alpha,
                              float num_boxes, const float *num_boxes_dev, float *out, float *comp, void *workspace,
                              int device, void *stream)
{
    if (int rc = pair_losses_args("mdetr_pair_losses_forward", L, B, Q, C, G, K)) return rc;
    if (!logits || !boxes || !dims || !depths || !angles || !assign || !labels || !boxes3d || !depth || !size3d ||
        !heading_bin || !heading_res || !valid || !num || !out || !comp || !workspace)
        return fail(MDETR_E_ARG, "mdetr_pair_losses_forward: null pointer");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_pair_losses_forward: set device %d: %s", device, hipGetErrorString(dev.err));
    const mdetr::PairLossDims d{L, B, Q, C, G, K, focal_alpha};
    const mdetr::PairLossIn in{logits, boxes, dims, depths, angles, assign, reinterpret_cast<const long long *>(labels),
                               reinterpret_cast<const long long *>(heading_bin), boxes3d, depth, size3d, heading_res, valid};
    const hipError_t e = mdetr::pair_losses_forward_launch(d, in, num, num_boxes, num_boxes_dev, out, comp, workspace,
                                                           static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_pair_losses_forward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
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
    if (int rc = pair_losses_args("mdetr_pair_losses_backward", L, B, Q, C, G, K)) return rc;
    if (!logits || !boxes || !dims || !depths || !angles || !assign || !labels || !boxes3d || !depth || !size3d ||
        !heading_bin || !heading_res || !valid || !grad_out || !comp || !g_logits || !g_boxes || !g_dims || !g_depths || !g_angles)
        return fail(MDETR_E_ARG, "mdetr_pair_losses_backward: null pointer");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_pair_losses_backward: set device %d: %s", device, hipGetErrorString(dev.err));
    const mdetr::PairLossDims d{L, B, Q, C, G, K, focal_alpha};
    const mdetr::PairLossIn in{logits, boxes, dims, depths, angles, assign, reinterpret_cast<const long long *>(labels),
                               reinterpret_cast<const long long *>(heading_bin), boxes3d, depth, size3d, heading_res, valid};
    const hipError_t e = mdetr::pair_losses_backward_launch(d, in, grad_out, comp, num_boxes, num_boxes_dev, g_logits,
                                                            g_boxes, g_dims, g_depths, g_angles,
                                                            static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_pair_losses_backward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_adamw_step(int param_dtype, void *param, float *master, const void *grad, float *exp_avg, float *exp_avg_sq,
                     int64_t n, int64_t n_no_decay, float beta1, float beta2, float eps, float weight_decay,
                     float step_size, const float *step_size_dev, int device, void *stream)
{
    if (param_dtype != MDETR_F32 && param_dtype != MDETR_BF16)
        return fail(MDETR_E_ARG, "mdetr_adamw_step: parameter dtype must be f32 or bf16");
    if (n < 0 || n_no_decay < 0 || n_no_decay > n) return fail(MDETR_E_ARG, "mdetr_adamw_step: bad sizes");
    if (n == 0) return MDETR_OK;
    if (!param || !master || !grad || !exp_avg || !exp_avg_sq) return fail(MDETR_E_ARG, "mdetr_adamw_step: null pointer");
    if (param_dtype == MDETR_F32 && static_cast<void *>(master) != param)
        return fail(MDETR_E_ARG, "mdetr_adamw_step: an f32 parameter is its own master copy");
    if (!aligned16(param) || !aligned16(master) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq))
        return fail(MDETR_E_ALIGN, "mdetr_adamw_step: buffers must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_adamw_step: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::adamw_launch(param_dtype, param, master, grad, exp_avg, exp_avg_sq, n, n_no_decay,
                                             beta1, beta2, eps, weight_decay, step_size, step_size_dev,
                                             static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_adamw_step: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_adamw_step_counted(int param_dtype, void *param, float *master, const void *grad, float *exp_avg, float *exp_avg_sq,
                             int64_t n, int64_t n_no_decay, float beta1, float beta2, float eps, float weight_decay,
                             const double *step_count_dev, float lr, const double *lr_dev, int device, void *stream)
{
    if (param_dtype != MDETR_F32 && param_dtype != MDETR_BF16)
        return fail(MDETR_E_ARG, "mdetr_adamw_step_counted: parameter dtype must be f32 or bf16");
    if (n < 0 || n_no_decay < 0 || n_no_decay > n) return fail(MDETR_E_ARG, "mdetr_adamw_step_counted: bad sizes");
    if (n == 0) return MDETR_OK;
    if (!param || !master || !grad || !exp_avg || !exp_avg_sq || !step_count_dev) return fail(MDETR_E_ARG, "mdetr_adamw_step_counted: null pointer");
    if (param_dtype == MDETR_F32 && static_cast<void *>(master) != param)
        return fail(MDETR_E_ARG, "mdetr_adamw_step_counted: an f32 parameter is its own master copy");
    if (!aligned16(param) || !aligned16(master) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq))
        return fail(MDETR_E_ALIGN, "mdetr_adamw_step_counted: buffers must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_adamw_step_counted: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::adamw_launch(param_dtype, param, master, grad, exp_avg, exp_avg_sq, n, n_no_decay, beta1, beta2, eps,
                                             weight_decay, 0.f, nullptr, static_cast<hipStream_t>(stream), step_count_dev, lr_dev, lr);
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_adamw_step_counted: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_adamw_step_gathered(int param_dtype, void *param, float *master, const void *const *grad_ptrs, int ntensors,
                              const int *tensor_block_begin, const int64_t *flat_offsets, const int64_t *nbytes, const int *block_tensor,
                              const int64_t *block_start, int chunk_bytes, float *exp_avg, float *exp_avg_sq, int64_t n_no_decay,
                              float beta1, float beta2, float eps, float weight_decay, float step_size, const float *step_size_dev,
                              const double *step_count_dev, float lr, const double *lr_dev, int device, void *stream)
{
    if (param_dtype != MDETR_F32 && param_dtype != MDETR_BF16)
        return fail(MDETR_E_ARG, "mdetr_adamw_step_gathered: parameter dtype must be f32 or bf16");
    const int esz = param_dtype == MDETR_F32 ? 4 : 2;
    if (ntensors < 0 || n_no_decay < 0 || chunk_bytes <= 0 || chunk_bytes % 16 != 0)
        return fail(MDETR_E_ARG, "mdetr_adamw_step_gathered: ntensors %d / chunk_bytes %d (a positive multiple of 16)", ntensors, chunk_bytes);
    if (ntensors == 0) return MDETR_OK;
    if (!param || !master || !grad_ptrs || !tensor_block_begin || !flat_offsets || !nbytes || !block_tensor || !block_start || !exp_avg || !exp_avg_sq)
        return fail(MDETR_E_ARG, "mdetr_adamw_step_gathered: null pointer");
    if (param_dtype == MDETR_F32 && static_cast<void *>(master) != param)
        return fail(MDETR_E_ARG, "mdetr_adamw_step_gathered: an f32 parameter is its own master copy");
    if (!aligned16(param) || !aligned16(master) || !aligned16(exp_avg) || !aligned16(exp_avg_sq))
        return fail(MDETR_E_ALIGN, "mdetr_adamw_step_gathered: buffers must be 16-byte aligned");
    for (int i = 0; i < ntensors; ++i)
        if (!grad_ptrs[i] || tensor_block_begin[i + 1] < tensor_block_begin[i] || (reinterpret_cast<uintptr_t>(grad_ptrs[i]) & (esz - 1)))
            return fail(MDETR_E_ARG, "mdetr_adamw_step_gathered: tensor %d: null / misaligned gradient or decreasing block table", i);
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_adamw_step_gathered: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::adamw_gathered_launch(param_dtype, param, master, grad_ptrs, ntensors, tensor_block_begin, flat_offsets, nbytes, block_tensor,
                                                      block_start, chunk_bytes, exp_avg, exp_avg_sq, n_no_decay, beta1, beta2, eps, weight_decay,
                                                      step_size, step_size_dev, static_cast<hipStream_t>(stream), step_count_dev, lr_dev, lr);
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_adamw_step_gathered: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_tgemm(const void *a, const void *w, const void *bias, const void *res, void *y, int64_t T, int N, int K,
                int64_t lda, int64_t ldw, int64_t ldr, int64_t ldy, int flags, float dropout_p, uint64_t seed,
                const void *seed_dev, int device, void *stream)
{
    if (T < 0 || N <= 0 || K <= 0) return fail(MDETR_E_ARG, "mdetr_tgemm: bad shape T=%lld N=%d K=%d", static_cast<long long>(T), N, K);
    if (T == 0) return MDETR_OK;
    if (!a || !w || !y) return fail(MDETR_E_ARG, "mdetr_tgemm: null pointer");
    if (flags & ~(MDETR_TGEMM_RELU | MDETR_TGEMM_NN | MDETR_TGEMM_BIAS_F32 | MDETR_TGEMM_OUT_F32)) return fail(MDETR_E_ARG, "mdetr_tgemm: unknown flag bits 0x%x", flags);
    if (dropout_p > 0.f && !(flags & MDETR_TGEMM_RELU)) return fail(MDETR_E_ARG, "mdetr_tgemm: dropout is only fused behind the ReLU");
    mdetr::TgemmProblem p{a, w, bias, res, y, T, N, K, lda, ldw, res ? ldr : 0, ldy, flags, dropout_p, seed, static_cast<const uint64_t *>(seed_dev)};
    if (!mdetr::tgemm_supported(p))
        return fail(MDETR_E_ARG, "mdetr_tgemm: needs bf16 operands, K %% 8 == 0, N %% 8 == 0, row strides %% 8 == 0 and >= the row length, 16-byte aligned "
                    "pointers, 0 <= dropout_p < 1 (T=%lld N=%d K=%d lda=%lld ldw=%lld ldr=%lld ldy=%lld flags=0x%x)", static_cast<long long>(T), N, K,
                    static_cast<long long>(lda), static_cast<long long>(ldw), static_cast<long long>(ldr), static_cast<long long>(ldy), flags);
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_tgemm: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::tgemm_launch(p, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_tgemm: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int64_t mdetr_sgemm_workspace_bytes(int mode, const mdetr_sgemm_problem *problems, int nprob)
{
    const char *why = mdetr::sgemm_check(mode, problems, nprob);
    if (why) return fail(MDETR_E_ARG, "mdetr_sgemm_workspace_bytes: %s (mode %d, %d problems)", why, mode, nprob);
    return mdetr::sgemm_workspace_bytes(mode, problems, nprob);
}

int mdetr_sgemm_grouped(int mode, const mdetr_sgemm_problem *problems, int nprob, void *workspace, int64_t workspace_bytes, int device, void *stream)
{
    const char *why = mdetr::sgemm_check(mode, problems, nprob);
    if (why) return fail(MDETR_E_ARG, "mdetr_sgemm_grouped: %s (mode %d, %d problems)", why, mode, nprob);
    const int64_t need = mdetr::sgemm_workspace_bytes(mode, problems, nprob);
    if (need > 0 && (!workspace || workspace_bytes < need || !aligned16(workspace)))
        return fail(MDETR_E_ARG, "mdetr_sgemm_grouped: needs a 16-byte aligned workspace of %lld bytes (mdetr_sgemm_workspace_bytes), got %lld",
                    static_cast<long long>(need), static_cast<long long>(workspace ? workspace_bytes : 0));
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_sgemm_grouped: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::sgemm_launch(mode, problems, nprob, workspace, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_sgemm_grouped: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_tgemm_masked(const void *a, const void *w, const void *res, const void *mask, void *y, int64_t T, int N, int K,
                       int64_t lda, int64_t ldw, int64_t ldr, int64_t ldm, int64_t ldy, int device, void *stream)
{
    if (T < 0 || N <= 0 || K <= 0) return fail(MDETR_E_ARG, "mdetr_tgemm_masked: bad shape T=%lld N=%d K=%d", static_cast<long long>(T), N, K);
    if (T == 0) return MDETR_OK;
    if (!a || !w || !y || !mask) return fail(MDETR_E_ARG, "mdetr_tgemm_masked: null pointer");
    mdetr::TgemmProblem p{a, w, nullptr, res, y, T, N, K, lda, ldw, res ? ldr : 0, ldy, MDETR_TGEMM_NN, 0.f, 0, nullptr};
    p.mask = mask; p.ldm = ldm;
    if (!mdetr::tgemm_supported(p))
        return fail(MDETR_E_ARG, "mdetr_tgemm_masked: needs bf16 operands, K %% 8 == 0, N %% 8 == 0, row strides %% 8 == 0 and >= the row length, 16-byte "
                    "aligned pointers (T=%lld N=%d K=%d lda=%lld ldw=%lld ldr=%lld ldm=%lld ldy=%lld)", static_cast<long long>(T), N, K,
                    static_cast<long long>(lda), static_cast<long long>(ldw), static_cast<long long>(ldr), static_cast<long long>(ldm), static_cast<long long>(ldy));
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_tgemm_masked: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::tgemm_launch(p, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_tgemm_masked: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int64_t mdetr_column_sum_workspace_bytes(int64_t rows, int cols)
{
    if (rows < 0 || cols < 0) return -1;
    return mdetr::colsum_workspace_bytes(rows, cols);
}

static int column_sum_impl(const char *who, int dtype, const void *x, void *out, int out_dtype, void *workspace, int64_t workspace_bytes,
                           int64_t rows, int cols, int64_t ld, int device, void *stream)
{
    if (rows < 0 || cols < 0 || ld < cols) return fail(MDETR_E_ARG, "%s: bad shape rows=%lld cols=%d ld=%lld", who,
                                                      static_cast<long long>(rows), cols, static_cast<long long>(ld));
    if (out_dtype != MDETR_F32 && out_dtype != MDETR_BF16) return fail(MDETR_E_ARG, "%s: out_dtype must be MDETR_F32 or MDETR_BF16", who);
    if (cols == 0) return MDETR_OK;
    if (!out || !workspace || (rows > 0 && !x)) return fail(MDETR_E_ARG, "%s: null pointer", who);
    if (!mdetr::colsum_supported(dtype, cols, ld, x))
        return fail(MDETR_E_ARG, "%s: needs f32 (cols %% 4 == 0) or bf16 (cols %% 8 == 0), 16-byte aligned rows", who);
    if (workspace_bytes < mdetr::colsum_workspace_bytes(rows, cols))
        return fail(MDETR_E_ARG, "%s: workspace too small", who);
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "%s: set device %d: %s", who, device, hipGetErrorString(dev.err));
    mdetr::ProfileScope prof(12, cols, static_cast<hipStream_t>(stream), 0.0, static_cast<double>(rows) * cols * (dtype == MDETR_BF16 ? 2.0 : 4.0) / 1e3);
    const hipError_t e = mdetr::colsum_launch(dtype, x, out, workspace, rows, cols, ld, static_cast<hipStream_t>(stream), out_dtype);
    if (e != hipSuccess) return fail(MDETR_E_HIP, "%s: launch failed: %s", who, hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_column_sum(int dtype, const void *x, float *out, void *workspace, int64_t workspace_bytes,
                     int64_t rows, int cols, int64_t ld, int device, void *stream)
{
    return column_sum_impl("mdetr_column_sum", dtype, x, out, MDETR_F32, workspace, workspace_bytes, rows, cols, ld, device, stream);
}

int mdetr_column_sum_to(int dtype, const void *x, void *out, int out_dtype, void *workspace, int64_t workspace_bytes,
                        int64_t rows, int cols, int64_t ld, int device, void *stream)
{
    return column_sum_impl("mdetr_column_sum_to", dtype, x, out, out_dtype, workspace, workspace_bytes, rows, cols, ld, device, stream);
}

int mdetr_box_refine(const float *delta, const float *ref, float *out, int64_t rows, int nd, int device, void *stream)
{
    if (rows < 0 || (nd != 2 && nd != 6)) return fail(MDETR_E_ARG, "mdetr_box_refine: bad sizes rows=%lld nd=%d (2 or 6)", static_cast<long long>(rows), nd);
    if (rows == 0) return MDETR_OK;
    if (!delta || !ref || !out) return fail(MDETR_E_ARG, "mdetr_box_refine: null pointer");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_box_refine: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::box_refine_launch(delta, ref, out, rows, nd, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_box_refine: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

static int head_tail_dims(const char *who, int L, int B, int Q, int nd0, int H, int W, mdetr::HeadTailDims *d)
{
    if (L <= 0 || B <= 0 || Q <= 0 || H <= 0 || W <= 0 || (nd0 != 2 && nd0 != 6) || static_cast<int64_t>(L) * B * Q * 6 >= (1ll << 31))
        return fail(MDETR_E_ARG, "%s: bad sizes L=%d B=%d Q=%d nd0=%d H=%d W=%d", who, L, B, Q, nd0, H, W);
    d->L = L; d->B = B; d->Q = Q; d->nd0 = nd0; d->H = H; d->W = W;
    return MDETR_OK;
}

int mdetr_head_tail_forward(const float *delta, const float *init_ref, const float *inter_refs, const float *size3d, const float *depth_reg,
                            const float *depth_map, const float *img_h, const float *focal, float *coord, float *depth_ave,
                            int L, int B, int Q, int nd0, int H, int W, int device, void *stream)
{
    mdetr::HeadTailDims d;
    if (int rc = head_tail_dims("mdetr_head_tail_forward", L, B, Q, nd0, H, W, &d)) return rc;
    if (!delta || !init_ref || (L > 1 && !inter_refs) || !size3d || !depth_reg || !depth_map || !img_h || !focal || !coord || !depth_ave)
        return fail(MDETR_E_ARG, "mdetr_head_tail_forward: null pointer");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_head_tail_forward: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::head_tail_forward_launch(d, delta, init_ref, inter_refs, size3d, depth_reg, depth_map, img_h, focal, coord, depth_ave,
                                                         static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_head_tail_forward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_head_tail_backward(const float *init_ref, const float *size3d, const float *depth_reg, const float *img_h, const float *focal,
                             const float *coord, const float *g_coord, const float *g_depth, float *g_delta, float *g_init_ref,
                             float *g_size3d, float *g_depth_reg, float *g_map, int L, int B, int Q, int nd0, int H, int W, int device, void *stream)
{
    mdetr::HeadTailDims d;
    if (int rc = head_tail_dims("mdetr_head_tail_backward", L, B, Q, nd0, H, W, &d)) return rc;
    if (!init_ref || !size3d || !depth_reg || !img_h || !focal || !coord || !g_delta || !g_init_ref || !g_size3d || !g_depth_reg)
        return fail(MDETR_E_ARG, "mdetr_head_tail_backward: null pointer");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_head_tail_backward: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::head_tail_backward_launch(d, init_ref, size3d, depth_reg, img_h, focal, coord, g_coord, g_depth, g_delta, g_init_ref,
                                                          g_size3d, g_depth_reg, g_map, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_head_tail_backward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_chunk_sums(const mdetr_chunk_job *jobs, int njobs, int device, void *stream)
{
    if (njobs == 0) return MDETR_OK;
    const char *why = mdetr::chunk_sums_check(jobs, njobs);
    if (why) return fail(MDETR_E_ARG, "mdetr_chunk_sums: %s (%d jobs)", why, njobs);
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_chunk_sums: set device %d: %s", device, hipGetErrorString(dev.err));
    double kb = 0.0;
    for (int i = 0; i < njobs; ++i) kb += (static_cast<double>(jobs[i].chunks) * jobs[i].cols * 4.0 + jobs[i].cols * (jobs[i].out_dtype == 2 ? 2.0 : 4.0)) / 1e3;
    mdetr::ProfileScope prof(12, njobs, static_cast<hipStream_t>(stream), 0.0, kb);
    const hipError_t e = mdetr::chunk_sums_launch(jobs, njobs, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_chunk_sums: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

static int conv3x3_any(const char *who, const void *x, const void *w, const float *shift, const void *mask, void *y, int B, int H, int W, int C, int N,
                       int relu, int device, void *stream)
{
    if (B < 0 || H < 0 || W < 0 || C <= 0 || N <= 0) return fail(MDETR_E_ARG, "%s: bad sizes B=%d H=%d W=%d C=%d N=%d", who, B, H, W, C, N);
    if (B == 0 || H == 0 || W == 0) return MDETR_OK;
    if (!x || !w || !y) return fail(MDETR_E_ARG, "%s: null pointer", who);
    if (!mdetr::conv3x3_supported(B, H, W, C, N, x, w, y) || (reinterpret_cast<uintptr_t>(mask) & 7) != 0)
        return fail(MDETR_E_ARG, "%s: needs C %% 64 == 0, N %% 32 == 0, 16-byte aligned x / w, 8-byte aligned y / mask (C=%d N=%d)", who, C, N);
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "%s: set device %d: %s", who, device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::conv3x3_launch(x, w, shift, y, B, H, W, C, N, (relu & 1) != 0, static_cast<hipStream_t>(stream), (relu & 2) != 0, mask);
    if (e != hipSuccess) return fail(MDETR_E_HIP, "%s: launch failed: %s", who, hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_conv3x3_plan(int B, int H, int W, int N)
{
    if (B <= 0 || H <= 0 || W <= 0 || N <= 0) return fail(MDETR_E_ARG, "mdetr_conv3x3_plan: bad sizes");
    return mdetr::conv3x3_plan(B, H, W, N);
}

int mdetr_conv3x3_forward(const void *x, const void *w, const float *shift, void *y, int B, int H, int W, int C, int N,
                          int relu, int device, void *stream)
{
    return conv3x3_any("mdetr_conv3x3_forward", x, w, shift, nullptr, y, B, H, W, C, N, relu, device, stream);
}

int mdetr_conv3x3_masked(const void *x, const void *w, const float *shift, const void *mask, void *y, int B, int H, int W, int C, int N,
                         int relu, int device, void *stream)
{
    if (!mask) return fail(MDETR_E_ARG, "mdetr_conv3x3_masked: null mask");
    return conv3x3_any("mdetr_conv3x3_masked", x, w, shift, mask, y, B, H, W, C, N, relu, device, stream);
}

int mdetr_conv_taps(const void *x, const void *w, const float *shift, void *y, const int64_t *dims, int relu, int device, void *stream)
{
    if (!dims) return fail(MDETR_E_ARG, "mdetr_conv_taps: null dims");
    mdetr::ConvTapsDims d;
    d.B = static_cast<int>(dims[0]); d.H = static_cast<int>(dims[1]); d.W = static_cast<int>(dims[2]); d.C = static_cast<int>(dims[3]);
    d.OH = static_cast<int>(dims[4]); d.OW = static_cast<int>(dims[5]); d.N = static_cast<int>(dims[6]);
    d.SI = static_cast<int>(dims[7]); d.TR = static_cast<int>(dims[8]); d.TS = static_cast<int>(dims[9]);
    d.PT = static_cast<int>(dims[10]); d.PL = static_cast<int>(dims[11]);
    d.ta0 = static_cast<int>(dims[12]); d.ta_step = static_cast<int>(dims[13]); d.te0 = static_cast<int>(dims[14]); d.te_step = static_cast<int>(dims[15]);
    d.y_off = dims[16]; d.y_sb = dims[17]; d.y_sr = dims[18]; d.y_sc = dims[19];
    d.w_sn = dims[20]; d.w_sa = dims[21]; d.w_se = dims[22];
    for (int i = 0; i < 12; ++i)
        if (dims[i] < 0 || dims[i] > (1ll << 30)) return fail(MDETR_E_ARG, "mdetr_conv_taps: bad size dims[%d] = %lld", i, static_cast<long long>(dims[i]));
    if (d.B == 0 || d.OH == 0 || d.OW == 0) return MDETR_OK;
    if (!x || !w || !y) return fail(MDETR_E_ARG, "mdetr_conv_taps: null pointer");
    if (!mdetr::conv_taps_supported(d, x, w, y))
        return fail(MDETR_E_ARG, "mdetr_conv_taps: unsupported problem (taps %dx%d stride %d, C=%d %% 64, N=%d %% 32, 16-byte aligned x / w, strides in whole "
                                 "8-byte units)", d.TR, d.TS, d.SI, d.C, d.N);
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_conv_taps: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::conv_taps_launch(x, w, shift, y, d, relu != 0, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_conv_taps: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_conv_taps_split(const void *x, const void *w, const float *shift, float *partial, int64_t partial_floats, const int64_t *dims,
                          int ksplit, int device, void *stream)
{
    if (!dims) return fail(MDETR_E_ARG, "mdetr_conv_taps_split: null dims");
    mdetr::ConvTapsDims d;
    d.B = static_cast<int>(dims[0]); d.H = static_cast<int>(dims[1]); d.W = static_cast<int>(dims[2]); d.C = static_cast<int>(dims[3]);
    d.OH = static_cast<int>(dims[4]); d.OW = static_cast<int>(dims[5]); d.N = static_cast<int>(dims[6]);
    d.SI = static_cast<int>(dims[7]); d.TR = static_cast<int>(dims[8]); d.TS = static_cast<int>(dims[9]);
    d.PT = static_cast<int>(dims[10]); d.PL = static_cast<int>(dims[11]);
    d.ta0 = static_cast<int>(dims[12]); d.ta_step = static_cast<int>(dims[13]); d.te0 = static_cast<int>(dims[14]); d.te_step = static_cast<int>(dims[15]);
    d.y_off = 0; d.y_sb = static_cast<int64_t>(d.OH) * d.OW * d.N; d.y_sr = static_cast<int64_t>(d.OW) * d.N; d.y_sc = d.N;
    d.w_sn = dims[20]; d.w_sa = dims[21]; d.w_se = dims[22];
    for (int i = 0; i < 12; ++i)
        if (dims[i] < 0 || dims[i] > (1ll << 30)) return fail(MDETR_E_ARG, "mdetr_conv_taps_split: bad size dims[%d] = %lld", i, static_cast<long long>(dims[i]));
    if (d.B == 0 || d.OH == 0 || d.OW == 0) return MDETR_OK;
    if (!x || !w || !partial) return fail(MDETR_E_ARG, "mdetr_conv_taps_split: null pointer");
    if (!mdetr::conv_taps_supported(d, x, w, partial) || !mdetr::conv_taps_split_supported(d, ksplit))
        return fail(MDETR_E_ARG, "mdetr_conv_taps_split: needs the 3x3 / stride-2 form, C a multiple of 32 x ksplit (C=%d ksplit=%d), 2 <= ksplit <= 64", d.C, ksplit);
    const int64_t need = static_cast<int64_t>(ksplit) * d.B * d.OH * d.OW * d.N;
    if (partial_floats < need) return fail(MDETR_E_ARG, "mdetr_conv_taps_split: partial buffer holds %lld floats, %lld needed", static_cast<long long>(partial_floats), static_cast<long long>(need));
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_conv_taps_split: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::conv_taps_split_launch(x, w, shift, partial, d, ksplit, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_conv_taps_split: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_conv_dgrad_s2(const void *dy, const void *wt, void *dx, int B, int OH, int OW, int N, int H, int W, int C, int K, int device, void *stream)
{
    if (B < 0 || OH < 0 || OW < 0 || H < 0 || W < 0 || N <= 0 || C <= 0) return fail(MDETR_E_ARG, "mdetr_conv_dgrad_s2: bad sizes");
    if (B == 0 || H == 0 || W == 0) return MDETR_OK;
    if (!dy || !wt || !dx) return fail(MDETR_E_ARG, "mdetr_conv_dgrad_s2: null pointer");
    if (!mdetr::conv_dgrad_s2_supported(B, OH, OW, N, H, W, C, K, dy, wt, dx))
        return fail(MDETR_E_ARG, "mdetr_conv_dgrad_s2: needs K in {1, 3}, N %% 64 == 0, C %% 32 == 0, OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1, 16-byte aligned "
                                 "dy / wt (K=%d N=%d C=%d %dx%d -> %dx%d)", K, N, C, H, W, OH, OW);
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_conv_dgrad_s2: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::conv_dgrad_s2_launch(dy, wt, dx, B, OH, OW, N, H, W, C, K, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_conv_dgrad_s2: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_conv_stem(const void *x, const void *w_packed, const float *shift, void *y, int B, int H, int W, int device, void *stream)
{
    if (B < 0 || H < 0 || W < 0) return fail(MDETR_E_ARG, "mdetr_conv_stem: bad sizes B=%d H=%d W=%d", B, H, W);
    if (B == 0 || H == 0 || W == 0) return MDETR_OK;
    if (!x || !w_packed || !y) return fail(MDETR_E_ARG, "mdetr_conv_stem: null pointer");
    if (!mdetr::conv_stem_supported(B, H, W, x, w_packed, y))
        return fail(MDETR_E_ARG, "mdetr_conv_stem: needs a 16-byte aligned packed weight, an 8-byte aligned output and an image batch below 2 GiB");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_conv_stem: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::conv_stem_launch(x, w_packed, shift, y, B, H, W, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_conv_stem: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

static mdetr::ConvWgradDims wgrad_dims(int B, int H, int W, int C, int OH, int OW, int N, int K, int SI)
{
    mdetr::ConvWgradDims d;
    d.B = B; d.H = H; d.W = W; d.C = C; d.OH = OH; d.OW = OW; d.N = N; d.K = K; d.SI = SI;
    return d;
}

int mdetr_conv_wgrad_chunks(int B, int H, int W, int C, int OH, int OW, int N, int K, int SI)
{
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || N <= 0) return 0;
    return mdetr::conv_wgrad_chunks(wgrad_dims(B, H, W, C, OH, OW, N, K, SI));
}

int mdetr_conv_wgrad(const void *x, const void *dy, float *partial, int64_t partial_floats, int B, int H, int W, int C, int OH, int OW, int N,
                     int K, int SI, int device, void *stream)
{
    if (B < 0 || H < 0 || W < 0 || C <= 0 || OH < 0 || OW < 0 || N <= 0)
        return fail(MDETR_E_ARG, "mdetr_conv_wgrad: bad sizes B=%d H=%d W=%d C=%d OH=%d OW=%d N=%d", B, H, W, C, OH, OW, N);
    if (B == 0 || OH == 0 || OW == 0) return fail(MDETR_E_ARG, "mdetr_conv_wgrad: empty problem (the caller zero-fills the gradient)");
    if (!x || !dy || !partial) return fail(MDETR_E_ARG, "mdetr_conv_wgrad: null pointer");
    const mdetr::ConvWgradDims d = wgrad_dims(B, H, W, C, OH, OW, N, K, SI);
    if (OH != (H + 2 * (K == 3 ? 1 : 0) - K) / SI + 1 || OW != (W + 2 * (K == 3 ? 1 : 0) - K) / SI + 1)
        return fail(MDETR_E_ARG, "mdetr_conv_wgrad: output map %dx%d does not belong to input %dx%d, %dx%d taps, stride %d", OH, OW, H, W, K, K, SI);
    if (!mdetr::conv_wgrad_supported(d, x, dy))
        return fail(MDETR_E_ARG, "mdetr_conv_wgrad: needs 3x3 (stride 1 / 2) or 1x1 (stride 2) taps, C %% 64 == 0, N %% 32 == 0, 16-byte aligned operands "
                                 "(K=%d SI=%d C=%d N=%d)", K, SI, C, N);
    const int64_t need = static_cast<int64_t>(mdetr::conv_wgrad_chunks(d)) * N * K * K * C;
    if (partial_floats < need) return fail(MDETR_E_ARG, "mdetr_conv_wgrad: partial buffer holds %lld floats, %lld needed", static_cast<long long>(partial_floats), static_cast<long long>(need));
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_conv_wgrad: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::conv_wgrad_launch(x, dy, partial, d, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_conv_wgrad: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

// csrc/twgrad.hip (transposing LDS reads) unless MDETR_TUNE="twgrad=0" asks for the 1x1 case of csrc/conv_wgrad.hip (tests)
static bool twgrad_wanted(int64_t T, int C, int N)
{
    char tune_buf[8];
    const char *ev = mdetr::tune_str("twgrad", tune_buf, sizeof(tune_buf));
    return !(ev && atoi(ev) == 0) && T > 0 && C > 0 && N > 0 && C % 8 == 0 && N % 8 == 0;
}

int mdetr_token_wgrad_chunks(int64_t T, int C, int N)
{
    if (twgrad_wanted(T, C, N)) return mdetr::twgrad_chunks(T, C, N);
    if (T <= 0 || T % 8 != 0 || T / 8 >= (1ll << 30) || C <= 0 || N <= 0) return 0;
    return mdetr::conv_wgrad_chunks(wgrad_dims(1, 8, static_cast<int>(T / 8), C, 8, static_cast<int>(T / 8), N, 1, 1));
}

int mdetr_token_wgrad(const void *x, const void *dy, float *partial, int64_t partial_floats, int64_t T, int C, int N, int with_bias,
                      int device, void *stream)
{
    if (!x || !dy || !partial) return fail(MDETR_E_ARG, "mdetr_token_wgrad: null pointer");
    if (twgrad_wanted(T, C, N)) {
        if (!mdetr::twgrad_supported(T, C, N, C, N, x, dy))
            return fail(MDETR_E_ARG, "mdetr_token_wgrad: needs C %% 8 == 0, N %% 8 == 0, 16-byte aligned operands below 2^31 bytes (T=%lld C=%d N=%d)",
                        static_cast<long long>(T), C, N);
        const int64_t need = static_cast<int64_t>(mdetr::twgrad_chunks(T, C, N)) * (static_cast<int64_t>(N) * C + (with_bias ? N : 0));
        if (partial_floats < need) return fail(MDETR_E_ARG, "mdetr_token_wgrad: partial buffer holds %lld floats, %lld needed", static_cast<long long>(partial_floats), static_cast<long long>(need));
        DeviceScope dev(device);
        if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_token_wgrad: set device %d: %s", device, hipGetErrorString(dev.err));
        const hipError_t e = mdetr::twgrad_launch(x, dy, partial, T, C, N, C, N, with_bias != 0, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_token_wgrad: launch failed: %s", hipGetErrorString(e));
        return MDETR_OK;
    }
    if (T <= 0 || T % 8 != 0 || T / 8 >= (1ll << 30) || C <= 0 || N <= 0)
        return fail(MDETR_E_ARG, "mdetr_token_wgrad: bad sizes T=%lld (a positive multiple of 8) C=%d N=%d", static_cast<long long>(T), C, N);
    // the [T, C] / [T, N] matrices as 1 x 8 x (T / 8) channels-last images: a 1x1 convolution's weight gradient
    mdetr::ConvWgradDims d = wgrad_dims(1, 8, static_cast<int>(T / 8), C, 8, static_cast<int>(T / 8), N, 1, 1);
    d.DB = with_bias ? 1 : 0;
    if (!mdetr::conv_wgrad_supported(d, x, dy))
        return fail(MDETR_E_ARG, "mdetr_token_wgrad: needs C %% 64 == 0, N %% 32 == 0, 16-byte aligned operands below 2^31 bytes (T=%lld C=%d N=%d)",
                    static_cast<long long>(T), C, N);
    const int64_t need = static_cast<int64_t>(mdetr::conv_wgrad_chunks(d)) * (static_cast<int64_t>(N) * C + (with_bias ? N : 0));
    if (partial_floats < need) return fail(MDETR_E_ARG, "mdetr_token_wgrad: partial buffer holds %lld floats, %lld needed", static_cast<long long>(partial_floats), static_cast<long long>(need));
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_token_wgrad: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::conv_wgrad_launch(x, dy, partial, d, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_token_wgrad: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_bias_act_forward(int io_dtype, int bias_dtype, const void *x, const void *bias, const void *skip, void *y,
                           int64_t rows, int cols, int relu, float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                           int device, void *stream)
{
    if (rows < 0) return fail(MDETR_E_ARG, "mdetr_bias_act_forward: negative row count");
    if (!mdetr::bias_act_supported(io_dtype, bias ? bias_dtype : MDETR_F32, cols))
        return fail(MDETR_E_ARG, "mdetr_bias_act_forward: io_dtype %d / bias_dtype %d / cols %d (f32: cols %% 4 == 0; bf16: cols %% 8 == 0, "
                                 "bf16 bias only with a bf16 activation)", io_dtype, bias_dtype, cols);
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail(MDETR_E_ARG, "mdetr_bias_act_forward: dropout_p = %g outside [0, 1)", static_cast<double>(dropout_p));
    if (rows == 0) return MDETR_OK;
    if (!x || !y) return fail(MDETR_E_ARG, "mdetr_bias_act_forward: null pointer");
    if (!aligned16(x) || !aligned16(y) || !aligned16(skip) || !aligned16(bias))
        return fail(MDETR_E_ALIGN, "mdetr_bias_act_forward: x, bias, skip, y must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_bias_act_forward: set device %d: %s", device, hipGetErrorString(dev.err));
    const mdetr::BiasActProblem p{io_dtype, bias ? bias_dtype : MDETR_F32, rows, cols, relu ? 1 : 0, dropout_p, seed, seed_dev};
    mdetr::ProfileScope prof(14, cols, static_cast<hipStream_t>(stream), 0.0, static_cast<double>(rows) * cols * (io_dtype == MDETR_BF16 ? 2.0 : 4.0) * (skip ? 3.0 : 2.0) / 1e3);
    const hipError_t e = mdetr::bias_act_forward_launch(p, x, bias, skip, y, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_bias_act_forward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_bias_act_backward(int io_dtype, const void *dy, const void *y, void *dx, int64_t rows, int cols, float scale,
                            int device, void *stream)
{
    if (rows < 0) return fail(MDETR_E_ARG, "mdetr_bias_act_backward: negative row count");
    if (!mdetr::bias_act_supported(io_dtype, MDETR_F32, cols))
        return fail(MDETR_E_ARG, "mdetr_bias_act_backward: io_dtype %d / cols %d (f32: cols %% 4 == 0; bf16: cols %% 8 == 0)", io_dtype, cols);
    if (rows == 0) return MDETR_OK;
    if (!dy || !y || !dx) return fail(MDETR_E_ARG, "mdetr_bias_act_backward: null pointer");
    if (!aligned16(dy) || !aligned16(y) || !aligned16(dx)) return fail(MDETR_E_ALIGN, "mdetr_bias_act_backward: dy, y, dx must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_bias_act_backward: set device %d: %s", device, hipGetErrorString(dev.err));
    mdetr::ProfileScope prof(14, cols, static_cast<hipStream_t>(stream), 0.0, static_cast<double>(rows) * cols * (io_dtype == MDETR_BF16 ? 2.0 : 4.0) * 3.0 / 1e3);
    const hipError_t e = mdetr::bias_act_backward_launch(io_dtype, dy, y, dx, rows, cols, scale, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_bias_act_backward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_decimate2(int backward, const void *src, void *dst, int B, int H, int W, int64_t pixel_bytes, int device, void *stream)
{
    const char *name = backward ? "mdetr_decimate2 (backward)" : "mdetr_decimate2";
    if (B < 0 || H < 0 || W < 0) return fail(MDETR_E_ARG, "%s: negative extent (B=%d H=%d W=%d)", name, B, H, W);
    if (B == 0 || H == 0 || W == 0) return MDETR_OK;
    if (!src || !dst) return fail(MDETR_E_ARG, "%s: null pointer", name);
    if (!mdetr::decimate2_supported(pixel_bytes, src, dst))
        return fail(MDETR_E_ALIGN, "%s: %lld bytes per pixel / pointers: both need 16-byte granularity", name, static_cast<long long>(pixel_bytes));
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "%s: set device %d: %s", name, device, hipGetErrorString(dev.err));
    const hipError_t e = backward ? mdetr::decimate2_backward_launch(src, dst, B, H, W, pixel_bytes, static_cast<hipStream_t>(stream))
                                  : mdetr::decimate2_forward_launch(src, dst, B, H, W, pixel_bytes, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "%s: launch failed: %s", name, hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_gather_flat(const void *const *src_ptrs, int ntensors, const int *tensor_block_begin, void *dst, const int64_t *dst_offsets,
                      const int64_t *nbytes, const int *block_tensor, const int64_t *block_start, int chunk_bytes, int device, void *stream)
{
    if (ntensors < 0 || chunk_bytes <= 0 || chunk_bytes % 16 != 0) return fail(MDETR_E_ARG, "mdetr_gather_flat: ntensors %d / chunk_bytes %d (a positive multiple of 16)", ntensors, chunk_bytes);
    if (ntensors == 0) return MDETR_OK;
    if (!src_ptrs || !tensor_block_begin || !dst || !dst_offsets || !nbytes || !block_tensor || !block_start) return fail(MDETR_E_ARG, "mdetr_gather_flat: null pointer");
    for (int i = 0; i < ntensors; ++i)
        if (!src_ptrs[i] || tensor_block_begin[i + 1] < tensor_block_begin[i]) return fail(MDETR_E_ARG, "mdetr_gather_flat: tensor %d: null source or decreasing block table", i);
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_gather_flat: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::gather_flat_launch(src_ptrs, ntensors, tensor_block_begin, dst, dst_offsets, nbytes, block_tensor, block_start, chunk_bytes,
                                                   static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_gather_flat: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

static int fold_args_ok(const char *who, int n, const void *const *a, const void *const *b, void *const *c, const int *O, const int *C, const int *taps)
{
    if (n < 0) return fail(MDETR_E_ARG, "%s: n = %d", who, n);
    if (n == 0) return MDETR_OK;
    if (!a || !b || !c || !O || !C || !taps) return fail(MDETR_E_ARG, "%s: null array", who);
    for (int i = 0; i < n; ++i) {
        if (!a[i] || !b[i] || !c[i]) return fail(MDETR_E_ARG, "%s: tensor %d: null pointer", who, i);
        if (!mdetr::fold_shape_supported(O[i], C[i], taps[i]))
            return fail(MDETR_E_ARG, "%s: tensor %d: O=%d C=%d taps=%d (O, C positive multiples of 8)", who, i, O[i], C[i], taps[i]);
        if (!aligned16(a[i]) || !aligned16(c[i]) || (reinterpret_cast<uintptr_t>(b[i]) & 3)) return fail(MDETR_E_ALIGN, "%s: tensor %d: 16-byte aligned tensors", who, i);
    }
    return MDETR_OK;
}

int mdetr_fold_weights(int n, const void *const *w, const void *const *scale, void *const *folded, void *const *folded_t,
                       const int *O, const int *C, const int *taps, int device, void *stream)
{
    const int rc = fold_args_ok("mdetr_fold_weights", n, w, scale, folded, O, C, taps);
    if (rc != MDETR_OK || n == 0) return rc;
    if (folded_t)
        for (int i = 0; i < n; ++i)
            if (folded_t[i] && !aligned16(folded_t[i])) return fail(MDETR_E_ALIGN, "mdetr_fold_weights: tensor %d: 16-byte aligned tensors", i);
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_fold_weights: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::fold_weights_launch(n, w, scale, folded, folded_t, O, C, taps, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_fold_weights: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_unfold_grads(int n, const void *const *dfolded, const void *const *scale, void *const *dw,
                       const int *O, const int *C, const int *taps, int device, void *stream)
{
    const int rc = fold_args_ok("mdetr_unfold_grads", n, dfolded, scale, dw, O, C, taps);
    if (rc != MDETR_OK || n == 0) return rc;
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_unfold_grads: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::unfold_grads_launch(n, dfolded, scale, dw, O, C, taps, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_unfold_grads: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_maxpool3x3s2_bf16(const void *x, void *y, int B, int H, int W, int C, int device, void *stream)
{
    if (B < 0 || H < 0 || W < 0 || C <= 0 || C % 8 != 0) return fail(MDETR_E_ARG, "mdetr_maxpool3x3s2_bf16: B=%d H=%d W=%d C=%d (C a positive multiple of 8)", B, H, W, C);
    if (B == 0 || H == 0 || W == 0) return MDETR_OK;
    if (!x || !y) return fail(MDETR_E_ARG, "mdetr_maxpool3x3s2_bf16: null pointer");
    if (!aligned16(x) || !aligned16(y)) return fail(MDETR_E_ALIGN, "mdetr_maxpool3x3s2_bf16: x, y must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_maxpool3x3s2_bf16: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::maxpool3x3s2_bf16_launch(x, y, B, H, W, C, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_maxpool3x3s2_bf16: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int64_t mdetr_small_wgrad_workspace_bytes(int64_t rows, int n, int k) { return mdetr::small_wgrad_workspace_bytes(rows, n, k); }

int mdetr_small_wgrad(int io_dtype, const void *dy, const void *x, void *out, int out_dtype, void *workspace, int64_t workspace_bytes,
                      int64_t rows, int n, int k, int64_t ldy, int64_t ldx, int device, void *stream)
{
    if (!mdetr::small_wgrad_supported(io_dtype, rows, n, k, ldy, ldx) || (out_dtype != MDETR_F32 && out_dtype != MDETR_BF16))
        return fail(MDETR_E_ARG, "mdetr_small_wgrad: io_dtype %d / out_dtype %d / rows %lld / n %d / k %d / ldy %lld / ldx %lld (f32 or bf16; 1 <= rows <= 8192, "
                                 "65536 for n <= 64; k a multiple of 64 with n * k <= 524288; ldx a multiple of 8 elements)", io_dtype, out_dtype,
                    static_cast<long long>(rows), n, k, static_cast<long long>(ldy), static_cast<long long>(ldx));
    if (!dy || !x || !out || !workspace) return fail(MDETR_E_ARG, "mdetr_small_wgrad: null pointer");
    if (workspace_bytes < mdetr::small_wgrad_workspace_bytes(rows, n, k))
        return fail(MDETR_E_ARG, "mdetr_small_wgrad: workspace of %lld bytes, need %lld", static_cast<long long>(workspace_bytes),
                    static_cast<long long>(mdetr::small_wgrad_workspace_bytes(rows, n, k)));
    if (!aligned16(x) || !aligned16(out) || !aligned16(workspace))
        return fail(MDETR_E_ALIGN, "mdetr_small_wgrad: x, out, workspace must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_small_wgrad: set device %d: %s", device, hipGetErrorString(dev.err));
    mdetr::ProfileScope prof(16, n, static_cast<hipStream_t>(stream), 2.0 * rows * n * k / 1e6, (static_cast<double>(rows) * (n + k) * (io_dtype == MDETR_BF16 ? 2.0 : 4.0) + 4.0 * n * k) / 1e3);
    const hipError_t e = mdetr::small_wgrad_launch(io_dtype, dy, x, out, workspace, rows, n, k, ldy, ldx, out_dtype, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_small_wgrad: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int64_t mdetr_group_norm_workspace_bytes(int n, int64_t hw, int c, int groups)
{
    return mdetr::group_norm_workspace_bytes(n, hw, c, groups);
}

static int group_norm_args(const char *fn, int io_dtype, int param_dtype, int n, int64_t hw, int c, int groups, int64_t workspace_bytes)
{
    if (n < 0 || hw < 0) return fail(MDETR_E_ARG, "%s: negative size", fn);
    if (!mdetr::group_norm_supported(io_dtype, param_dtype, c, groups) || (io_dtype == MDETR_F32 && param_dtype == MDETR_BF16))
        return fail(MDETR_E_ARG, "%s: io_dtype %d / param_dtype %d / c %d / groups %d (f32 or bf16; bf16 parameters only with a bf16 activation; "
                                 "c == 8 * groups, c / 8 a power of two <= 256)", fn, io_dtype, param_dtype, c, groups);
    if (n > 0 && hw > 0 && workspace_bytes < mdetr::group_norm_workspace_bytes(n, hw, c, groups))
        return fail(MDETR_E_ARG, "%s: workspace of %lld bytes, need %lld", fn, static_cast<long long>(workspace_bytes),
                    static_cast<long long>(mdetr::group_norm_workspace_bytes(n, hw, c, groups)));
    return MDETR_OK;
}

int mdetr_group_norm_forward(int io_dtype, int param_dtype, const void *x, const void *gamma, const void *beta, void *y, float *stats,
                             void *workspace, int64_t workspace_bytes, int n, int64_t hw, int c, int groups, float eps, int relu,
                             int device, void *stream)
{
    if (int rc = group_norm_args("mdetr_group_norm_forward", io_dtype, param_dtype, n, hw, c, groups, workspace_bytes)) return rc;
    if (n == 0 || hw == 0) return MDETR_OK;
    if (!x || !gamma || !beta || !y || !stats || !workspace) return fail(MDETR_E_ARG, "mdetr_group_norm_forward: null pointer");
    if (!aligned16(x) || !aligned16(y) || !aligned16(gamma) || !aligned16(beta) || !aligned16(workspace))
        return fail(MDETR_E_ALIGN, "mdetr_group_norm_forward: x, y, gamma, beta, workspace must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_group_norm_forward: set device %d: %s", device, hipGetErrorString(dev.err));
    const mdetr::GroupNormProblem p{io_dtype, param_dtype, n, hw, c, groups, eps, relu ? 1 : 0};
    mdetr::ProfileScope prof(15, c, static_cast<hipStream_t>(stream), 0.0, 2.0 * n * hw * c * (io_dtype == MDETR_BF16 ? 2.0 : 4.0) / 1e3);
    const hipError_t e = mdetr::group_norm_forward_launch(p, x, gamma, beta, y, stats, workspace, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_group_norm_forward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_group_norm_backward(int io_dtype, int param_dtype, const void *dy, const void *x, const void *gamma, const void *beta,
                              const float *stats, void *dx, void *dparams, void *workspace, int64_t workspace_bytes,
                              int n, int64_t hw, int c, int groups, int relu, int device, void *stream)
{
    if (int rc = group_norm_args("mdetr_group_norm_backward", io_dtype, param_dtype, n, hw, c, groups, workspace_bytes)) return rc;
    if (!dparams) return fail(MDETR_E_ARG, "mdetr_group_norm_backward: null dparams");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_group_norm_backward: set device %d: %s", device, hipGetErrorString(dev.err));
    if (n == 0 || hw == 0) {
        const hipError_t e = mdetr::zero_fill_launch(dparams, static_cast<int64_t>(2) * c * (param_dtype == MDETR_BF16 ? 2 : 4), static_cast<hipStream_t>(stream));
        return e == hipSuccess ? MDETR_OK : fail(MDETR_E_HIP, "mdetr_group_norm_backward: zero fill failed: %s", hipGetErrorString(e));
    }
    if (!dy || !x || !gamma || !beta || !stats || !dx || !workspace) return fail(MDETR_E_ARG, "mdetr_group_norm_backward: null pointer");
    if (!aligned16(dy) || !aligned16(x) || !aligned16(dx) || !aligned16(gamma) || !aligned16(beta) || !aligned16(dparams) || !aligned16(workspace))
        return fail(MDETR_E_ALIGN, "mdetr_group_norm_backward: dy, x, dx, gamma, beta, dparams, workspace must be 16-byte aligned");
    const mdetr::GroupNormProblem p{io_dtype, param_dtype, n, hw, c, groups, 0.f, relu ? 1 : 0};
    mdetr::ProfileScope prof(15, c, static_cast<hipStream_t>(stream), 0.0, 3.0 * n * hw * c * (io_dtype == MDETR_BF16 ? 2.0 : 4.0) / 1e3);
    const hipError_t e = mdetr::group_norm_backward_launch(p, dy, x, gamma, beta, stats, dx, dparams, workspace, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_group_norm_backward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

static int add_ln_args(const char *fn, int io_dtype, int param_dtype, int64_t rows, int cols, float dropout_p)
{
    if (io_dtype != MDETR_F32 && io_dtype != MDETR_BF16) return fail(MDETR_E_ARG, "%s: io_dtype %d (MDETR_F32 or MDETR_BF16)", fn, io_dtype);
    if (param_dtype != MDETR_F32 && param_dtype != MDETR_BF16) return fail(MDETR_E_ARG, "%s: param_dtype %d (MDETR_F32 or MDETR_BF16)", fn, param_dtype);
    if (rows < 0) return fail(MDETR_E_ARG, "%s: negative row count", fn);
    if (cols != 128 && cols != 256 && cols != 512) return fail(MDETR_E_ARG, "%s: cols = %d (128, 256 or 512)", fn, cols);
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail(MDETR_E_ARG, "%s: dropout_p = %g outside [0, 1)", fn, static_cast<double>(dropout_p));
    return MDETR_OK;
}

int mdetr_add_layernorm_forward(int io_dtype, int param_dtype, const void *a, const void *b, const void *gamma, const void *beta, void *y,
                                void *s, float *stats, int64_t rows, int cols, float eps, float dropout_p, uint64_t seed,
                                const uint64_t *seed_dev, int device, void *stream)
{
    if (int rc = add_ln_args("mdetr_add_layernorm_forward", io_dtype, param_dtype, rows, cols, dropout_p)) return rc;
    if (rows == 0) return MDETR_OK;
    if (!a || !gamma || !beta || !y || !stats) return fail(MDETR_E_ARG, "mdetr_add_layernorm_forward: null pointer");
    if (!aligned16(a) || !aligned16(b) || !aligned16(y) || !aligned16(s))
        return fail(MDETR_E_ALIGN, "mdetr_add_layernorm_forward: a, b, y, s must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_add_layernorm_forward: set device %d: %s", device, hipGetErrorString(dev.err));
    const mdetr::AddLnProblem p{io_dtype, param_dtype, rows, cols, eps, dropout_p, seed, seed_dev};
    mdetr::ProfileScope prof(13, cols, static_cast<hipStream_t>(stream), 0.0, 4.0 * rows * cols * (io_dtype == MDETR_BF16 ? 2.0 : 4.0) / 1e3);
    const hipError_t e = mdetr::add_ln_forward_launch(p, a, b, gamma, beta, y, s, stats, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_add_layernorm_forward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int64_t mdetr_add_layernorm_partial_rows(int64_t rows) { return mdetr::add_ln_partial_rows(rows); }

int mdetr_add_layernorm_backward(int io_dtype, int param_dtype, const void *dy, const void *s, const void *gamma, const float *stats,
                                 void *da, void *db, float *partial, int64_t rows, int cols, float dropout_p,
                                 uint64_t seed, const uint64_t *seed_dev, int device, void *stream)
{
    if (int rc = add_ln_args("mdetr_add_layernorm_backward", io_dtype, param_dtype, rows, cols, dropout_p)) return rc;
    if (!partial) return fail(MDETR_E_ARG, "mdetr_add_layernorm_backward: null partial buffer");
    if (rows > 0 && (!dy || !s || !gamma || !stats || !da)) return fail(MDETR_E_ARG, "mdetr_add_layernorm_backward: null pointer");
    if (!aligned16(dy) || !aligned16(s) || !aligned16(da) || !aligned16(db))
        return fail(MDETR_E_ALIGN, "mdetr_add_layernorm_backward: dy, s, da, db must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_add_layernorm_backward: set device %d: %s", device, hipGetErrorString(dev.err));
    const mdetr::AddLnProblem p{io_dtype, param_dtype, rows, cols, 0.f, dropout_p, seed, seed_dev};
    hipError_t e = hipSuccess;
    if (rows == 0) {
        e = mdetr::zero_fill_launch(partial, static_cast<int64_t>(2) * cols * 4, static_cast<hipStream_t>(stream));
    } else {
        mdetr::ProfileScope prof(13, cols, static_cast<hipStream_t>(stream), 0.0, 4.0 * rows * cols * (io_dtype == MDETR_BF16 ? 2.0 : 4.0) / 1e3);
        e = mdetr::add_ln_backward_launch(p, dy, s, gamma, stats, da, db, partial, static_cast<hipStream_t>(stream));
    }
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_add_layernorm_backward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

static int riou_args(const char *fn, const void *boxes, const void *qboxes, const int64_t *box_start, const int64_t *qbox_start,
                     const int64_t *out_start, int n_frames, int64_t total_pairs, int criterion, const void *out)
{
    if (n_frames < 0 || total_pairs < 0) return fail(MDETR_E_ARG, "%s: negative size", fn);
    if (criterion < -1 || criterion > 2) return fail(MDETR_E_ARG, "%s: criterion %d (expected -1, 0, 1 or 2)", fn, criterion);
    if (total_pairs > 0 && (!boxes || !qboxes || !box_start || !qbox_start || !out_start || !out)) return fail(MDETR_E_ARG, "%s: null pointer", fn);
    return MDETR_OK;
}

int mdetr_rotate_iou_eval(const float *boxes, const float *qboxes, const int64_t *box_start, const int64_t *qbox_start,
                          const int64_t *out_start, int n_frames, int64_t total_pairs, int criterion, float *out,
                          int device, void *stream)
{
    if (int rc = riou_args("mdetr_rotate_iou_eval", boxes, qboxes, box_start, qbox_start, out_start, n_frames, total_pairs, criterion, out)) return rc;
    if (total_pairs == 0) return MDETR_OK;
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_rotate_iou_eval: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::rotate_iou_launch(boxes, qboxes, box_start, qbox_start, out_start, n_frames, total_pairs, criterion, out,
                                                  static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_rotate_iou_eval: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_box3d_overlap_eval(const double *boxes, const double *qboxes, const int64_t *box_start, const int64_t *qbox_start,
                             const int64_t *out_start, int n_frames, int64_t total_pairs, int criterion, double *out,
                             int device, void *stream)
{
    if (int rc = riou_args("mdetr_box3d_overlap_eval", boxes, qboxes, box_start, qbox_start, out_start, n_frames, total_pairs, criterion, out)) return rc;
    if (total_pairs == 0) return MDETR_OK;
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_box3d_overlap_eval: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::box3d_overlap_launch(boxes, qboxes, box_start, qbox_start, out_start, n_frames, total_pairs, criterion, out,
                                                     static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_box3d_overlap_eval: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_kitti_preprocess(const uint8_t *pixels, const MdetrKittiImage *images, int n_images, void *out,
                           int out_dtype, int out_h, int out_w, int channels_last, const float *mean, const float *std,
                           int device, void *stream)
{
    if (n_images < 0 || out_h < 0 || out_w < 0) return fail(MDETR_E_ARG, "mdetr_kitti_preprocess: negative size");
    if (out_dtype != MDETR_F32 && out_dtype != MDETR_BF16)
        return fail(MDETR_E_ARG, "mdetr_kitti_preprocess: out_dtype %d (MDETR_F32 or MDETR_BF16)", out_dtype);
    if (out_w % 4 != 0) return fail(MDETR_E_ARG, "mdetr_kitti_preprocess: out_w = %d must be a multiple of 4", out_w);
    if (!mean || !std) return fail(MDETR_E_ARG, "mdetr_kitti_preprocess: null mean / std");
    if (n_images == 0 || out_h == 0 || out_w == 0) return MDETR_OK;
    if (!pixels || !images || !out) return fail(MDETR_E_ARG, "mdetr_kitti_preprocess: null pointer");
    if (!aligned16(out) || reinterpret_cast<uintptr_t>(images) % 8 != 0)
        return fail(MDETR_E_ALIGN, "mdetr_kitti_preprocess: out must be 16-byte, images 8-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_kitti_preprocess: set device %d: %s", device, hipGetErrorString(dev.err));
    mdetr::KittiNorm norm;
    for (int c = 0; c < 3; ++c) { norm.mean[c] = mean[c]; norm.stdv[c] = std[c]; }
    const hipError_t e = mdetr::kitti_prep_launch(pixels, images, n_images, out, out_dtype, out_h, out_w, norm, channels_last != 0,
                                                  static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_kitti_preprocess: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_msda_forward_bf16(const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                            const float *loc, const float *attn, void *out,
                            int B, int S, int M, int D, int L, int Lq, int P, int device, void *stream)
{
    if (int rc = check_common("mdetr_msda_forward_bf16", MDETR_F32, B, S, M, D, L, Lq, P)) return rc;
    if (D != 32 || L != 4 || P != 4) return fail(MDETR_E_ARG, "mdetr_msda_forward_bf16: D = 32, L = P = 4 only (D=%d L=%d P=%d)", D, L, P);
    if (B == 0 || Lq == 0) return MDETR_OK;
    if (!value || !spatial_shapes || !level_start || !loc || !attn || !out)
        return fail(MDETR_E_ARG, "mdetr_msda_forward_bf16: null pointer");
    if (!aligned16(value) || !aligned16(loc) || !aligned16(attn) || !aligned16(out))
        return fail(MDETR_E_ALIGN, "mdetr_msda_forward_bf16: value/loc/attn/out must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_forward_bf16: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::msda_forward_bf16_launch(value, spatial_shapes, level_start, loc, attn, out,
                                                         B, S, M, D, L, Lq, P, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_forward_bf16: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_msda_backward_bf16(const void *value, const int64_t *spatial_shapes, const int64_t *level_start,
                             const float *loc, const float *attn, const void *grad_out,
                             float *grad_value, float *grad_loc, float *grad_attn,
                             int B, int S, int M, int D, int L, int Lq, int P,
                             const int64_t *spatial_shapes_host, const int64_t *level_start_host,
                             void *workspace, int64_t workspace_bytes, int device, void *stream)
{
    if (int rc = check_common("mdetr_msda_backward_bf16", MDETR_F32, B, S, M, D, L, Lq, P)) return rc;
    if (D != 32 || L != 4 || P != 4) return fail(MDETR_E_ARG, "mdetr_msda_backward_bf16: D = 32, L = P = 4 only (D=%d L=%d P=%d)", D, L, P);
    if (!grad_value || !grad_loc || !grad_attn) return fail(MDETR_E_ARG, "mdetr_msda_backward_bf16: null output pointer");
    if (B > 0 && Lq > 0 && (!value || !spatial_shapes || !level_start || !loc || !attn || !grad_out))
        return fail(MDETR_E_ARG, "mdetr_msda_backward_bf16: null pointer");
    if (!aligned16(value) || !aligned16(loc) || !aligned16(attn) || !aligned16(grad_out) || !aligned16(grad_value) ||
        !aligned16(grad_loc) || !aligned16(grad_attn))
        return fail(MDETR_E_ALIGN, "mdetr_msda_backward_bf16: all tensors must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_backward_bf16: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::msda_backward_bf16_launch(value, spatial_shapes, level_start, loc, attn, grad_out, grad_value,
                                                          grad_loc, grad_attn, B, S, M, D, L, Lq, P, spatial_shapes_host,
                                                          level_start_host, workspace, workspace_bytes,
                                                          static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_backward_bf16: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

static int prologue_args(const char *who, int io_dtype, int B, int Lq, int M, int L, int P, int R)
{
    if (io_dtype != MDETR_F32 && io_dtype != MDETR_BF16) return fail(MDETR_E_ARG, "%s: dtype must be f32 or bf16", who);
    if (B < 0 || Lq < 0 || M <= 0 || L <= 0 || P <= 0 || L * P > mdetr::kPrologueMaxLP || (R != 2 && R != 6))
        return fail(MDETR_E_ARG, "%s: bad shape B=%d Lq=%d M=%d L=%d P=%d R=%d", who, B, Lq, M, L, P, R);
    return MDETR_OK;
}

int mdetr_msda_prologue_forward(int io_dtype, int ref_dtype, const void *offsets, const void *logits, const void *ref,
                                const int64_t *spatial_shapes, float *sampling_loc, float *attn_weight,
                                int B, int Lq, int M, int L, int P, int R, int64_t ref_sb, int64_t ref_sq, int64_t ref_sl,
                                int device, void *stream)
{
    if (int rc = prologue_args("mdetr_msda_prologue_forward", io_dtype, B, Lq, M, L, P, R)) return rc;
    if (ref_dtype != MDETR_F32 && ref_dtype != MDETR_BF16) return fail(MDETR_E_ARG, "mdetr_msda_prologue_forward: ref_dtype %d", ref_dtype);
    if (B == 0 || Lq == 0) return MDETR_OK;
    if (!offsets || !logits || !ref || !spatial_shapes || !sampling_loc || !attn_weight)
        return fail(MDETR_E_ARG, "mdetr_msda_prologue_forward: null pointer");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_prologue_forward: set device %d: %s", device, hipGetErrorString(dev.err));
    const mdetr::PrologueDims d{B, Lq, M, L, P, R, ref_sb, ref_sq, ref_sl};
    const hipError_t e = mdetr::msda_prologue_forward_launch(io_dtype, ref_dtype, d, offsets, logits, ref, spatial_shapes, sampling_loc,
                                                             attn_weight, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_prologue_forward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_msda_prologue_backward(int io_dtype, int ref_dtype, const void *offsets, const void *ref, const int64_t *spatial_shapes,
                                 const float *attn_weight, const float *grad_loc, const float *grad_attn,
                                 void *grad_offsets, void *grad_logits, float *grad_ref,
                                 int B, int Lq, int M, int L, int P, int R, int64_t ref_sb, int64_t ref_sq, int64_t ref_sl,
                                 int device, void *stream)
{
    if (int rc = prologue_args("mdetr_msda_prologue_backward", io_dtype, B, Lq, M, L, P, R)) return rc;
    if (ref_dtype != MDETR_F32 && ref_dtype != MDETR_BF16) return fail(MDETR_E_ARG, "mdetr_msda_prologue_backward: ref_dtype %d", ref_dtype);
    if (B == 0 || Lq == 0) return MDETR_OK;
    if (!offsets || !ref || !spatial_shapes || !attn_weight || !grad_loc || !grad_attn || !grad_offsets || !grad_logits)
        return fail(MDETR_E_ARG, "mdetr_msda_prologue_backward: null pointer");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_prologue_backward: set device %d: %s", device, hipGetErrorString(dev.err));
    const mdetr::PrologueDims d{B, Lq, M, L, P, R, ref_sb, ref_sq, ref_sl};
    hipError_t e = hipSuccess;
    if (grad_ref) e = mdetr::zero_fill_launch(grad_ref, static_cast<int64_t>(B) * Lq * L * R * 4, static_cast<hipStream_t>(stream));
    if (e == hipSuccess)
        e = mdetr::msda_prologue_backward_launch(io_dtype, ref_dtype, d, offsets, ref, spatial_shapes, attn_weight, grad_loc, grad_attn,
                                                 grad_offsets, grad_logits, grad_ref, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_prologue_backward: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

// the packed form: ONE projection output [B, Lq, M L P 3] holding, per query, the M L P 2 offsets and then the M L P logits
int mdetr_msda_prologue_forward_packed(int io_dtype, int ref_dtype, const void *packed, const void *ref, const int64_t *spatial_shapes,
                                       float *sampling_loc, float *attn_weight, int B, int Lq, int M, int L, int P, int R,
                                       int64_t ref_sb, int64_t ref_sq, int64_t ref_sl, int device, void *stream)
{
    if (int rc = prologue_args("mdetr_msda_prologue_forward_packed", io_dtype, B, Lq, M, L, P, R)) return rc;
    if (ref_dtype != MDETR_F32 && ref_dtype != MDETR_BF16) return fail(MDETR_E_ARG, "mdetr_msda_prologue_forward_packed: ref_dtype %d", ref_dtype);
    if (B == 0 || Lq == 0) return MDETR_OK;
    if (!packed || !ref || !spatial_shapes || !sampling_loc || !attn_weight) return fail(MDETR_E_ARG, "mdetr_msda_prologue_forward_packed: null pointer");
    if (L != 4 || P != 4) return fail(MDETR_E_ARG, "mdetr_msda_prologue_forward_packed: the packed form needs L = P = 4 (L=%d P=%d)", L, P);
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_prologue_forward_packed: set device %d: %s", device, hipGetErrorString(dev.err));
    mdetr::PrologueDims d{B, Lq, M, L, P, R, ref_sb, ref_sq, ref_sl};
    d.po = d.pl = static_cast<int64_t>(M) * L * P * 3;
    const size_t esz = io_dtype == MDETR_BF16 ? 2 : 4;
    const void *logits = static_cast<const char *>(packed) + static_cast<size_t>(M) * L * P * 2 * esz;
    const hipError_t e = mdetr::msda_prologue_forward_launch(io_dtype, ref_dtype, d, packed, logits, ref, spatial_shapes, sampling_loc,
                                                             attn_weight, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_prologue_forward_packed: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_msda_prologue_backward_packed(int io_dtype, int ref_dtype, const void *packed, const void *ref, const int64_t *spatial_shapes,
                                        const float *attn_weight, const float *grad_loc, const float *grad_attn, void *grad_packed,
                                        float *grad_ref, int B, int Lq, int M, int L, int P, int R,
                                        int64_t ref_sb, int64_t ref_sq, int64_t ref_sl, int device, void *stream)
{
    if (int rc = prologue_args("mdetr_msda_prologue_backward_packed", io_dtype, B, Lq, M, L, P, R)) return rc;
    if (ref_dtype != MDETR_F32 && ref_dtype != MDETR_BF16) return fail(MDETR_E_ARG, "mdetr_msda_prologue_backward_packed: ref_dtype %d", ref_dtype);
    if (B == 0 || Lq == 0) return MDETR_OK;
    if (!packed || !ref || !spatial_shapes || !attn_weight || !grad_loc || !grad_attn || !grad_packed)
        return fail(MDETR_E_ARG, "mdetr_msda_prologue_backward_packed: null pointer");
    if (L != 4 || P != 4) return fail(MDETR_E_ARG, "mdetr_msda_prologue_backward_packed: the packed form needs L = P = 4 (L=%d P=%d)", L, P);
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_prologue_backward_packed: set device %d: %s", device, hipGetErrorString(dev.err));
    mdetr::PrologueDims d{B, Lq, M, L, P, R, ref_sb, ref_sq, ref_sl};
    d.po = d.pl = static_cast<int64_t>(M) * L * P * 3;
    const size_t esz = io_dtype == MDETR_BF16 ? 2 : 4;
    void *g_logits = static_cast<char *>(grad_packed) + static_cast<size_t>(M) * L * P * 2 * esz;
    hipError_t e = hipSuccess;
    if (grad_ref) e = mdetr::zero_fill_launch(grad_ref, static_cast<int64_t>(B) * Lq * L * R * 4, static_cast<hipStream_t>(stream));
    if (e == hipSuccess)
        e = mdetr::msda_prologue_backward_launch(io_dtype, ref_dtype, d, packed, ref, spatial_shapes, attn_weight, grad_loc, grad_attn,
                                                 grad_packed, g_logits, grad_ref, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_prologue_backward_packed: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_lsa_forward_fused(const float *logits, const float *boxes, const int64_t *labels, const float *boxes3d,
                            const int32_t *num_targets, int32_t *assign, int layers, int images, int groups, int n,
                            int kmax, int num_classes, float w_class, float w_bbox, float w_center, float w_giou,
                            float focal_alpha, int device, void *stream)
{
    if (layers < 0 || images < 0 || groups < 0 || n <= 0 || n > 128 || kmax < 0 || kmax > n || kmax > 64 || num_classes <= 0)
        return fail(MDETR_E_ARG, "mdetr_lsa_forward_fused: need 0 < n <= 128, 0 <= kmax <= min(n, 64), num_classes > 0 (n=%d kmax=%d)", n, kmax);
    if (layers == 0 || images == 0 || groups == 0 || kmax == 0) return MDETR_OK;
    if (!logits || !boxes || !labels || !boxes3d || !num_targets || !assign)
        return fail(MDETR_E_ARG, "mdetr_lsa_forward_fused: null pointer");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_lsa_forward_fused: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::lsa_fused_launch(logits, boxes, labels, boxes3d, num_targets, assign, layers, images, groups,
                                                 n, kmax, num_classes, w_class, w_bbox, w_center, w_giou, focal_alpha,
                                                 static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_lsa_forward_fused: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

int mdetr_msda_indices(int dtype, const int64_t *spatial_shapes, const void *loc, int32_t *idx,
                       int B, int M, int L, int Lq, int P, int device, void *stream)
{
    if (int rc = check_common("mdetr_msda_indices", dtype, B, 0, M, 1, L, Lq, P)) return rc;
    if (B == 0 || Lq == 0) return MDETR_OK;
    if (!spatial_shapes || !loc || !idx) return fail(MDETR_E_ARG, "mdetr_msda_indices: null pointer");
    if (!aligned16(idx)) return fail(MDETR_E_ALIGN, "mdetr_msda_indices: idx must be 16-byte aligned");
    DeviceScope dev(device);
    if (dev.err != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_indices: set device %d: %s", device, hipGetErrorString(dev.err));
    const hipError_t e = mdetr::msda_indices_launch(dtype, spatial_shapes, loc, idx, B, M, L, Lq, P,
                                                    static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(MDETR_E_HIP, "mdetr_msda_indices: launch failed: %s", hipGetErrorString(e));
    return MDETR_OK;
}

}  // extern "C"
