// monodetr_amd/csrc/msda_cpu.hip -- host (CPU) implementation of the MSDA operator behind the C ABI's
// mdetr_msda_forward_cpu / mdetr_msda_backward_cpu.
//
// The reference's CPU entry points only raise (ops/src/cpu/ms_deform_attn_cpu.cpp:17-40: "Not implement on cpu"), which is
// why its BASELINE configs[0] ("configs/monodetr.yaml on CPU, 1 train iteration, plumbing") cannot run at all.  These
// two functions follow the arithmetic of the reference's CUDA kernels (ms_deform_im2col_cuda.cuh:33-84 / 237-299 forward,
// :87-159 / 301-403 backward) in plain C++: one (image, head) per task on a small pool of std::threads -- heads own
// disjoint channels of grad_value, so the scatter needs no atomics and is deterministic.
//
// They are EXPLICIT entry points for host tensors, not a fallback: the GPU entry points never route here, and
// monodetr_amd/msda_ext.py keeps the reference's behaviour (CPU tensors raise) unless allow_cpu(True) asks for them.
// Host-only code: nothing here runs on the device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "msda.h"

namespace mdetr {
namespace {

template <typename T>
struct Tap {                       // one bilinear footprint
    bool inside;                   // .cuh:288
    int y0, x0;
    T ly, lx, hy, hx;
    bool ok[4];                    // (y0,x0) (y0,x0+1) (y0+1,x0) (y0+1,x0+1) inside the map (.cuh:56-74)
};

template <typename T>
inline Tap<T> make_tap(T loc_x, T loc_y, int H, int W)
{
    Tap<T> t;
    volatile T py = loc_y * static_cast<T>(H);      // product rounded before the subtraction (.cuh:285-286; no FMA)
    volatile T px = loc_x * static_cast<T>(W);
    const T h = py - static_cast<T>(0.5), w = px - static_cast<T>(0.5);
    t.inside = h > T(-1) && w > T(-1) && h < static_cast<T>(H) && w < static_cast<T>(W);
    const T hs = t.inside ? h : T(0), ws = t.inside ? w : T(0);
    const T fy = std::floor(hs), fx = std::floor(ws);
    t.y0 = static_cast<int>(fy);
    t.x0 = static_cast<int>(fx);
    t.ly = hs - fy; t.lx = ws - fx; t.hy = T(1) - t.ly; t.hx = T(1) - t.lx;
    const bool yl = t.y0 >= 0, xl = t.x0 >= 0, yh = t.y0 + 1 <= H - 1, xh = t.x0 + 1 <= W - 1;
    t.ok[0] = t.inside && yl && xl; t.ok[1] = t.inside && yl && xh;
    t.ok[2] = t.inside && yh && xl; t.ok[3] = t.inside && yh && xh;
    return t;
}

template <typename F>
void for_each_image_head(int B, int M, F body)
{
    const int tasks = B * M;
    int nt = static_cast<int>(std::thread::hardware_concurrency());
    nt = std::max(1, std::min(nt, tasks));
    if (nt == 1) { for (int t = 0; t < tasks; ++t) body(t / M, t % M); return; }
    std::vector<std::thread> pool;
    for (int k = 0; k < nt; ++k)
        pool.emplace_back([=] { for (int t = k; t < tasks; t += nt) body(t / M, t % M); });
    for (auto &th : pool) th.join();
}

template <typename T>
void forward_cpu(const T *value, const int64_t *shapes, const int64_t *lstart, const T *loc, const T *attn, T *out,
                 int B, int S, int M, int D, int L, int Lq, int P)
{
    const int64_t row = static_cast<int64_t>(M) * D;
    for_each_image_head(B, M, [=](int b, int m) {
        for (int q = 0; q < Lq; ++q) {
            const int64_t pair = (static_cast<int64_t>(b) * Lq + q) * M + m;
            T *o = out + pair * D;
            for (int c = 0; c < D; ++c) o[c] = T(0);
            for (int l = 0; l < L; ++l) {
                const int H = static_cast<int>(shapes[2 * l]), W = static_cast<int>(shapes[2 * l + 1]);
                const T *vl = value + (static_cast<int64_t>(b) * S + lstart[l]) * row + static_cast<int64_t>(m) * D;
                for (int p = 0; p < P; ++p) {
                    const int64_t s = (pair * L + l) * P + p;
                    const Tap<T> t = make_tap(loc[2 * s], loc[2 * s + 1], H, W);
                    if (!t.inside) continue;
                    const T a = attn[s];
                    const T w4[4] = {t.hy * t.hx, t.hy * t.lx, t.ly * t.hx, t.ly * t.lx};
                    const T *v4[4];
                    for (int k = 0; k < 4; ++k)
                        v4[k] = vl + (static_cast<int64_t>(t.y0 + (k >> 1)) * W + t.x0 + (k & 1)) * row;
                    for (int c = 0; c < D; ++c) {
                        T val = T(0);                                          // .cuh:80-82
                        for (int k = 0; k < 4; ++k) if (t.ok[k]) val += w4[k] * v4[k][c];
                        o[c] += a * val;                                       // .cuh:290
                    }
                }
            }
        }
    });
}

template <typename T>
void backward_cpu(const T *value, const int64_t *shapes, const int64_t *lstart, const T *loc, const T *attn, const T *go,
                  T *gv, T *gl, T *ga, int B, int S, int M, int D, int L, int Lq, int P)
{
    const int64_t row = static_cast<int64_t>(M) * D;
    for_each_image_head(B, M, [=](int b, int m) {
        for (int64_t pix = 0; pix < S; ++pix) {                               // this (image, head)'s slice of grad_value
            T *z = gv + (static_cast<int64_t>(b) * S + pix) * row + static_cast<int64_t>(m) * D;
            for (int c = 0; c < D; ++c) z[c] = T(0);
        }
        for (int q = 0; q < Lq; ++q) {
            const int64_t pair = (static_cast<int64_t>(b) * Lq + q) * M + m;
            const T *g = go + pair * D;
            for (int l = 0; l < L; ++l) {
                const int H = static_cast<int>(shapes[2 * l]), W = static_cast<int>(shapes[2 * l + 1]);
                const int64_t base = (static_cast<int64_t>(b) * S + lstart[l]) * row + static_cast<int64_t>(m) * D;
                for (int p = 0; p < P; ++p) {
                    const int64_t s = (pair * L + l) * P + p;
                    const Tap<T> t = make_tap(loc[2 * s], loc[2 * s + 1], H, W);
                    T dx = T(0), dy = T(0), da = T(0);                        // .cuh:365-367: zero outside the window
                    if (t.inside) {
                        const T a = attn[s];
                        const T w4[4] = {t.hy * t.hx, t.hy * t.lx, t.ly * t.hx, t.ly * t.lx};
                        int64_t off[4];
                        for (int k = 0; k < 4; ++k)
                            off[k] = base + (static_cast<int64_t>(t.y0 + (k >> 1)) * W + t.x0 + (k & 1)) * row;
                        for (int c = 0; c < D; ++c) {
                            const T top = g[c] * a;                           // .cuh:113
                            T v[4] = {T(0), T(0), T(0), T(0)};
                            for (int k = 0; k < 4; ++k)
                                if (t.ok[k]) { v[k] = value[off[k] + c]; gv[off[k] + c] += w4[k] * top; }   // .cuh:125-152
                            const T gh = -t.hx * v[0] - t.lx * v[1] + t.hx * v[2] + t.lx * v[3];
                            const T gw = -t.hy * v[0] + t.hy * v[1] - t.ly * v[2] + t.ly * v[3];
                            da += g[c] * (w4[0] * v[0] + w4[1] * v[1] + w4[2] * v[2] + w4[3] * v[3]);      // .cuh:156
                            dx += static_cast<T>(W) * gw * top;                                             // .cuh:157
                            dy += static_cast<T>(H) * gh * top;                                             // .cuh:158
                        }
                    }
                    gl[2 * s] = dx; gl[2 * s + 1] = dy; ga[s] = da;
                }
            }
        }
    });
}

}  // namespace

// dtype 0 = f32, 1 = f64; every pointer is a HOST pointer (shapes and level starts too)
void msda_forward_cpu(int dtype, const void *value, const int64_t *shapes, const int64_t *lstart, const void *loc,
                      const void *attn, void *out, int B, int S, int M, int D, int L, int Lq, int P)
{
    if (dtype == 0) forward_cpu(static_cast<const float *>(value), shapes, lstart, static_cast<const float *>(loc), static_cast<const float *>(attn), static_cast<float *>(out), B, S, M, D, L, Lq, P);
    else forward_cpu(static_cast<const double *>(value), shapes, lstart, static_cast<const double *>(loc), static_cast<const double *>(attn), static_cast<double *>(out), B, S, M, D, L, Lq, P);
}

void msda_backward_cpu(int dtype, const void *value, const int64_t *shapes, const int64_t *lstart, const void *loc,
                       const void *attn, const void *grad_out, void *grad_value, void *grad_loc, void *grad_attn,
                       int B, int S, int M, int D, int L, int Lq, int P)
{
    if (dtype == 0)
        backward_cpu(static_cast<const float *>(value), shapes, lstart, static_cast<const float *>(loc), static_cast<const float *>(attn), static_cast<const float *>(grad_out),
                     static_cast<float *>(grad_value), static_cast<float *>(grad_loc), static_cast<float *>(grad_attn), B, S, M, D, L, Lq, P);
    else
        backward_cpu(static_cast<const double *>(value), shapes, lstart, static_cast<const double *>(loc), static_cast<const double *>(attn), static_cast<const double *>(grad_out),
                     static_cast<double *>(grad_value), static_cast<double *>(grad_loc), static_cast<double *>(grad_attn), B, S, M, D, L, Lq, P);
}

}  // namespace mdetr
