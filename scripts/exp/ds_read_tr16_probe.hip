#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short *out, int mode)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;
    __syncthreads();
    const int l = threadIdx.x;
    int addr;   // element index of this lane's 4 contiguous values
    if (mode == 0) addr = (l & 15) * 4 + (l >> 4) * 64;            // simple: lane l reads elements 4l..4l+3
    else if (mode == 1) addr = (l & 15) * 64 + (l >> 4) * 4;       // rows of 64 elements: lane (l&15) row, 4 cols at (l>>4)*4
    else addr = (l & 3) * 64 + ((l >> 2) & 3) * 4 + (l >> 4) * 256;
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3))) *)(lds + addr));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}
int main()
{
    unsigned short *d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    }
    return 0;
}
