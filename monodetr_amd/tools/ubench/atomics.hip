// Developer micro-benchmark: fp32 atomic-add throughput on gfx950 by access pattern.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomics.hip -o atomics && ./atomics
// Patterns: (a) streaming contiguous global atomics, (b) random 128-B rows (what msda_bwd does),
// (c) random rows confined to a small hot set (contention), (d) LDS ds_add_f32, (e) plain stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_stream_atomic(float *dst, size_t n, int passes)
{
    for (int p = 0; p < passes; ++p)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
            unsafeAtomicAdd(dst + i, 1.0f);
}

__global__ void k_stream_store(float *dst, size_t n, int passes)
{
    for (int p = 0; p < passes; ++p)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
            dst[i] = (float)p;
}

// each half-wave adds to one random 128-B row (32 floats)
__global__ void k_row_atomic(float *dst, const unsigned *rows, size_t nrows_total, int per_thread)
{
    const size_t hw = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 5;
    const int c = threadIdx.x & 31;
    const size_t nhw = ((size_t)gridDim.x * blockDim.x) >> 5;
    for (int k = 0; k < per_thread; ++k) {
        const unsigned r = rows[(hw + (size_t)k * nhw) % nrows_total];
        unsafeAtomicAdd(dst + (size_t)r * 32 + c, 1.0f);
    }
}

// MODE 0: ds_add_f32, MODE 1: plain ds read-modify-write (non-atomic, for the rate ceiling),
// MODE 2: ds_add_u32, MODE 3: ds_add_u64
template <int MODE>
__global__ void k_lds_atomic(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    unsigned s = (threadIdx.x >> 5) * 2654435761u + blockIdx.x * 40503u + 17u;    // shared by a half-wave
    const int c = threadIdx.x & 31;
    for (int k = 0; k < iters; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            const unsigned row = (s >> 8) & 511;
            if (MODE == 0) __hip_atomic_fetch_add(&lds[row * 32 + c], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 1) lds[row * 32 + c] += 1.0f;
            else if (MODE == 2) __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(lds) + row * 32 + c, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(__builtin_assume_aligned(lds, 8)) + (row & 255) * 32 + c, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[5];
}

template <typename F>
float timeit(F f, int reps = 5)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const size_t n = 8ull * 10200 * 256;        // grad_value of the encoder call: 20.9 M floats
    float *dst; CK(hipMalloc(&dst, n * 4)); CK(hipMemset(dst, 0, n * 4));
    const int passes = 4;
    float ms = timeit([&] { k_stream_atomic<<<2048, 256>>>(dst, n, passes); });
    printf("stream atomic : %.3f ms  %.1f G atomics/s  (%.2f TB/s payload)\n", ms, n * passes / ms / 1e6, n * passes * 4 / ms / 1e9);
    ms = timeit([&] { k_stream_store<<<2048, 256>>>(dst, n, passes); });
    printf("stream store  : %.3f ms  %.2f TB/s\n", ms, n * passes * 4 / ms / 1e9);

    const size_t nrows = n / 32;
    for (size_t hot : {nrows, (size_t)65536, (size_t)4096, (size_t)256}) {
        std::vector<unsigned> h(1 << 22);
        unsigned s = 12345;
        for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 4) % hot; }
        unsigned *rows; CK(hipMalloc(&rows, h.size() * 4)); CK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        const int per = 64, blocks = 4096;
        const double total = (double)blocks * 256 * per;
        ms = timeit([&] { k_row_atomic<<<blocks, 256>>>(dst, rows, h.size(), per); });
        printf("row atomic (hot set %8zu rows): %.3f ms  %.1f G atomics/s\n", hot, ms, total / ms / 1e6);
        CK(hipFree(rows));
    }
    float *out; CK(hipMalloc(&out, 4096 * 4));
    const int iters = 4096;
    for (int thr : {256, 512, 1024}) {
        ms = timeit([&] { k_lds_atomic<0><<<512, thr, 65536>>>(out, iters); });
        printf("lds ds_add_f32 (512 blocks x %4d thr): %.3f ms  %.1f G atomics/s  (%.2f per clk per CU @2.1GHz)\n", thr, ms, 512.0 * thr * iters / ms / 1e6, 512.0 * thr * iters / ms / 1e6 / 256 / 2.1);
        ms = timeit([&] { k_lds_atomic<1><<<512, thr, 65536>>>(out, iters); });
        printf("lds plain rmw  (512 blocks x %4d thr): %.3f ms  %.1f G ops/s\n", thr, ms, 512.0 * thr * iters / ms / 1e6);
        ms = timeit([&] { k_lds_atomic<2><<<512, thr, 65536>>>(out, iters); });
        printf("lds ds_add_u32 (512 blocks x %4d thr): %.3f ms  %.1f G atomics/s\n", thr, ms, 512.0 * thr * iters / ms / 1e6);
        ms = timeit([&] { k_lds_atomic<3><<<512, thr, 65536>>>(out, iters); });
        printf("lds ds_add_u64 (512 blocks x %4d thr): %.3f ms  %.1f G atomics/s\n", thr, ms, 512.0 * thr * iters / ms / 1e6);
    }
    return 0;
}
