// How fast does v_mfma_f32_32x32x2_f32 issue on gfx950?  One wave per SIMD (256 CUs x 4 waves), N instructions per wave:
//   chain = 1: every instruction depends on the previous one's accumulator; chain = 4: four independent accumulators round-robin.
// Prints cycles per instruction per SIMD from the wall clock (at whatever clock the chip runs) and from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-mfma-vgpr-form] -o /tmp/mfma_f32_rate scripts/exp/mfma_f32_rate.hip && /tmp/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int CHAIN>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int n, float a, float b)
{
    f32x16 acc[CHAIN];
    for (int c = 0; c < CHAIN; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = threadIdx.x * 1e-3f + c;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i += CHAIN) {
#pragma unroll
        for (int c = 0; c < CHAIN; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < CHAIN; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAIN> void run(int blocks, int n)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<CHAIN>, dim3(blocks), dim3(256), 0, 0, out, cyc, n, 1.0001f, 0.9999f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
        if (rep == 2)
            printf("chain %d blocks %d: %.1f us for %d instructions per wave -> %.1f ns each, s_memtime %.1f ticks each, %.1f TFLOP/s\n", CHAIN, blocks,
                   ms * 1e3, n, ms * 1e6 / n, (double)c0 / n, 4096.0 * n * blocks * 4 / (ms * 1e-3) / 1e12);
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<1>(256, 20000); run<4>(256, 20000); run<1>(512, 20000); run<1>(1024, 20000);
    return 0;
}
