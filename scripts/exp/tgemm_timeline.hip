// Clock marks inside tgemm_kernel (csrc/tgemm.hip compiled here with MDETR_TGEMM_TIMELINE) for one small product: where a
// workgroup's time goes when the launch is ~one workgroup per CU.   scripts/exp/tgemm_timeline.sh <tag> T K N
#define MDETR_TGEMM_TIMELINE 1
#include "../../monodetr_amd/csrc/tgemm.hip"

#include <stdio.h>
#include <vector>

int main(int argc, char **argv)
{
    const int T = argc > 1 ? atoi(argv[1]) : 4400, K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 256;
    std::vector<unsigned short> ha(static_cast<size_t>(T) * K, 0x3c00), hw(static_cast<size_t>(N) * K, 0x3c00), hb(N, 0x3c00);
    void *a, *w, *b, *y;
    (void)hipMalloc(&a, ha.size() * 2); (void)hipMalloc(&w, hw.size() * 2); (void)hipMalloc(&b, N * 2); (void)hipMalloc(&y, static_cast<size_t>(T) * N * 2);
    (void)hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, hb.data(), N * 2, hipMemcpyHostToDevice);
    mdetr::TgemmProblem p{};
    p.a = a; p.w = w; p.bias = b; p.res = nullptr; p.y = y; p.T = T; p.N = N; p.K = K; p.lda = K; p.ldw = K; p.ldr = 0; p.ldy = N; p.flags = 0;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 6; ++it) {
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < 10; ++r) (void)mdetr::tgemm_launch(p, 0);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("10 launches: %.1f us each\n", ms * 100);
    }
    long long tl[64];
    (void)hipMemcpyFromSymbol(tl, HIP_SYMBOL(mdetr::tgemm_tl), sizeof(tl));
    printf("workgroup 3, thread 0 (clock ticks from kernel entry): args staged %lld, first fetches issued %lld, slab 0 deposited %lld (the wait), barrier %lld\n",
           tl[1] - tl[0], tl[2] - tl[0], tl[3] - tl[0], tl[4] - tl[0]);
    for (int s = 0; s < K / 64 && s < 10; ++s)
        printf("  slab %d: deposit+fetch %lld, products %lld, barrier %lld\n", s, tl[5 + 3 * s] - (s ? tl[7 + 3 * (s - 1)] : tl[4]), tl[6 + 3 * s] - tl[5 + 3 * s], tl[7 + 3 * s] - tl[6 + 3 * s]);
    printf("  park %lld, barrier %lld, drain %lld;  total %lld ticks\n", tl[40] - tl[7 + 3 * (K / 64 - 1)], tl[41] - tl[40], tl[42] - tl[41], tl[42] - tl[0]);
    return 0;
}
