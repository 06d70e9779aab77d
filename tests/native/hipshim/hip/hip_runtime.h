// tests/native/hipshim/hip/hip_runtime.h -- TEST INFRASTRUCTURE.  A minimal "HIP on the CPU" so that the repository's
// own .hip sources (kernels, launchers and the C ABI in capi.hip) compile UNCHANGED with g++ and run block by block
// on the host: it shadows <hip/hip_runtime.h> when tests/native_emul.py puts this directory first on the include
// path.  Purpose: exercise the launch glue of kernels that have not met a GPU yet -- grid / block geometry, index
// arithmetic, LDS reductions, last-block finalisation, workspace handling, argument checks -- with the same ctypes
// wrappers the product uses.  What it cannot show: data races, memory-model issues, performance.
//
// Execution model (runtime.cpp): the threads of a block are fibers on ONE OS thread, run in lane order up to
// their next synchronisation point; __syncthreads releases when every live thread of the block waits, wave-level
// exchanges (__shfl*) when every live lane of the 64-wide wave waits.  Blocks run one after another; "atomics"
// are plain read-modify-writes.  Only what the covered sources use is provided.
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotSupported = 801 };
typedef void *hipStream_t;
typedef void *hipEvent_t;

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern dim3 threadIdx, blockIdx, blockDim, gridDim;
constexpr int warpSize = 64;

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline int2 make_int2(int x, int y) { return {x, y}; }
inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __expf(float x) { return expf(x); }
inline float __logf(float x) { return logf(x); }
inline float __fsqrt_rn(float x) { return sqrtf(x); }
inline float __fdividef(float a, float b) { return a / b; }

namespace hipshim {
void run(dim3 grid, dim3 block, size_t lds, const std::function<void()> &body);
void sync_block();
void sync_wave();
extern unsigned long long exchange[1024];     // one 8-byte slot per thread of the block
int lane_base();                              // first thread index of the calling thread's wave
int live_lanes();                             // threads of this wave that exist in the block
}  // namespace hipshim

inline void __syncthreads() { hipshim::sync_block(); }
inline unsigned __umul24(unsigned a, unsigned b) { return static_cast<unsigned>(static_cast<unsigned long long>(a & 0xFFFFFFu) * (b & 0xFFFFFFu)); }   // v_mul_u32_u24
inline void __threadfence() {}
inline void __threadfence_block() {}

template <typename T> inline T __shfl(T v, int src_lane, int width = 64)
{
    static_assert(sizeof(T) <= 8, "shuffle of at most 8 bytes");
    const int me = threadIdx.x, base = hipshim::lane_base();
    memcpy(&hipshim::exchange[me], &v, sizeof(T));
    hipshim::sync_wave();
    const int lane = me - base, seg = lane / width * width;
    T r;
    memcpy(&r, &hipshim::exchange[base + seg + (src_lane % width)], sizeof(T));
    hipshim::sync_wave();
    return r;
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64)
{
    const int lane = static_cast<int>(threadIdx.x) - hipshim::lane_base();
    return __shfl(v, (lane % width) ^ mask, width);
}
template <typename T> inline T __shfl_down(T v, int delta, int width = 64)
{
    const int lane = static_cast<int>(threadIdx.x) - hipshim::lane_base();
    const int src = (lane % width) + delta;
    return __shfl(v, src < width ? src : lane % width, width);
}

// wave votes: every live lane of the wave takes part
inline int __any(int pred)
{
    const int me = threadIdx.x, base = hipshim::lane_base(), n = hipshim::live_lanes();
    hipshim::exchange[me] = pred != 0;
    hipshim::sync_wave();
    int r = 0;
    for (int l = 0; l < n; ++l) r |= static_cast<int>(hipshim::exchange[base + l]);
    hipshim::sync_wave();
    return r;
}
inline int __all(int pred) { return !__any(!pred); }
// 64-bit mask of the live lanes whose predicate holds
inline unsigned long long __ballot(int pred)
{
    const int me = threadIdx.x, base = hipshim::lane_base(), n = hipshim::live_lanes();
    hipshim::exchange[me] = pred != 0;
    hipshim::sync_wave();
    unsigned long long r = 0;
    for (int l = 0; l < n; ++l) r |= (hipshim::exchange[base + l] ? 1ull : 0ull) << l;
    hipshim::sync_wave();
    return r;
}
// block-wide vote with a barrier on both sides
inline int __syncthreads_or(int pred)
{
    hipshim::exchange[threadIdx.x] = pred != 0;
    hipshim::sync_block();
    int r = 0;
    for (unsigned t = 0; t < blockDim.x * blockDim.y * blockDim.z; ++t) r |= static_cast<int>(hipshim::exchange[t]);
    hipshim::sync_block();
    return r;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

template <typename T, typename U> inline T atomicAdd(T *p, U v) { const T old = *p; *p = old + static_cast<T>(v); return old; }
template <typename T, typename U> inline T unsafeAtomicAdd(T *p, U v) { return atomicAdd(p, v); }
template <typename T> inline T atomicMax(T *p, T v) { const T old = *p; if (v > old) *p = v; return old; }

template <typename T, typename U> inline T __hip_atomic_fetch_add(T *p, U v, int, int) { return atomicAdd(p, v); }
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
using std::max;
using std::min;

// ---- AMDGCN builtins used by the covered sources ------------------------------------------------------------------
inline void __builtin_amdgcn_fence(int, const char *) {}
inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}               // a hint to the instruction scheduler: nothing to emulate
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_wave_barrier() { hipshim::sync_wave(); }        // lanes run one after another here: a real rendezvous
inline void __builtin_amdgcn_s_barrier() { hipshim::sync_block(); }
// value of the wave's first live lane (every live lane of the wave takes part)
inline int __builtin_amdgcn_readfirstlane(int v)
{
    const int me = threadIdx.x, base = hipshim::lane_base();
    memcpy(&hipshim::exchange[me], &v, sizeof(int));
    hipshim::sync_wave();
    int r;
    memcpy(&r, &hipshim::exchange[base], sizeof(int));
    hipshim::sync_wave();
    return r;
}
inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }                // only used on wave-uniform values
// v_mov_b32 with a DPP modifier, all rows / banks enabled, bound_ctrl: quad_perm, row_shl / row_shr, row_mirror,
// row_half_mirror (the controls the sources use); a row is 16 lanes.  Invalid source lanes read 0 (bound_ctrl).
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    (void)old; (void)row_mask; (void)bank_mask; (void)bound_ctrl;
    const int lane = static_cast<int>(threadIdx.x) - hipshim::lane_base();
    int from = -1;
    if (ctrl <= 0xFF) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int s = (lane & 15) + (ctrl & 15); from = s < 16 ? (lane & ~15) + s : -1; }
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int s = (lane & 15) - (ctrl & 15); from = s >= 0 ? (lane & ~15) + s : -1; }
    else if (ctrl == 0x140) from = (lane & ~15) | (15 - (lane & 15));
    else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));
    else { fprintf(stderr, "hipshim: DPP control 0x%x not emulated\n", ctrl); abort(); }
    const int got = __shfl(src, from < 0 ? lane : from);          // every lane takes part in the exchange
    return from < 0 ? 0 : got;
}

template <typename K, typename... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t, A... args)
{
    hipshim::run(grid, block, lds, [&]() { kernel(args...); });
}

hipError_t hipGetLastError();
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
hipError_t hipFuncSetAttribute(const void *, int, int);
const char *hipGetErrorString(hipError_t);
hipError_t hipGetDevice(int *);
hipError_t hipSetDevice(int);
hipError_t hipEventCreate(hipEvent_t *);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventElapsedTime(float *, hipEvent_t, hipEvent_t);
