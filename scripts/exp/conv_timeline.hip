// Clock marks inside conv3x3_kernel (csrc/conv3x3.hip compiled here with MDETR_CONV3X3_TIMELINE): where a workgroup's stages spend
// their time.    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I include -I monodetr_amd/csrc \
//                    scripts/exp/conv_timeline.hip -L monodetr_amd -lmonodetr_amd -Wl,-rpath,$PWD/monodetr_amd -o /tmp/conv_timeline
//                /tmp/conv_timeline C H W            (B = 8, N = C)
#define MDETR_CONV3X3_TIMELINE 1
#include "../../monodetr_amd/csrc/conv3x3.hip"

#include <stdio.h>
#include <algorithm>
#include <vector>

int main(int argc, char **argv)
{
    const int C = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 24, W = argc > 3 ? atoi(argv[3]) : 80, B = 8, N = C;
    const size_t nx = static_cast<size_t>(B) * H * W * C, nw = static_cast<size_t>(N) * 9 * C, ny = static_cast<size_t>(B) * H * W * N;
    std::vector<unsigned short> hx(nx), hw(nw);
    for (size_t i = 0; i < nx; ++i) hx[i] = 0x3c00 + (i * 2654435761u >> 24);            // bf16 bit patterns near 0.01
    for (size_t i = 0; i < nw; ++i) hw[i] = 0x3c00 + (i * 40503u >> 9 & 0xff);
    void *x, *w, *y;
    hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&y, ny * 2);
    hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0, 0);
        mdetr::conv3x3_launch(x, w, nullptr, y, B, H, W, C, N, true, 0, false);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.1f us\n", it, ms * 1e3);
    }
    static long long tl[16][4][128];
    hipMemcpyFromSymbol(tl, HIP_SYMBOL(mdetr::conv_tl), sizeof(tl));
    const int slabs = C / 64;
    long long t00 = tl[0][0][0];
    for (int s = 0; s < 16; s += 3) {
        printf("workgroup slot %d (block %d)\n", s, s * 67);
        for (int wv = 0; wv < 4; ++wv) {
            const long long *t = tl[s][wv];
            printf("  wave %d: start %+lld  total %lld ticks | per slab [wait-barrier store+fetch barrier products] x 3 stages:\n", wv, t[0] - t00, t[127] - t[0]);
            for (int sl = 0; sl < slabs && sl < 8; ++sl) {
                const long long *q = t + 1 + sl * 15;
                printf("    slab %d:", sl);
                for (int st = 0; st < 3; ++st) printf("  [%lld %lld %lld %lld]", q[st * 4 + 1] - q[st * 4], q[st * 4 + 2] - q[st * 4 + 1], q[st * 4 + 3] - q[st * 4 + 2], q[st * 4 + 4] - q[st * 4 + 3]);
                printf("\n");
            }
        }
    }
    // occupancy over time on one XCD (its workgroups share a clock): starts and ends of every workgroup
    static long long span[4096][4];
    hipMemcpyFromSymbol(span, HIP_SYMBOL(mdetr::conv_span), sizeof(span));
    const int tiles = B * ((H + 3) / 4) * ((W + 31) / 32);
    int nblk = 0;
    for (int i = 0; i < 4096; ++i) if (span[i][1] > span[i][0] && span[i][0] != 0) nblk = i + 1;
    {   // the 100 MHz wall clock is one clock for the whole device
        std::vector<std::pair<long long, int>> ev;
        long long t0 = -1, t1 = 0; int n = 0;
        for (int i = 0; i < nblk; ++i) if (span[i][1] > span[i][0]) {
            if (t0 < 0 || span[i][0] < t0) t0 = span[i][0];
            if (span[i][1] > t1) t1 = span[i][1];
            ev.push_back({span[i][0], 1}); ev.push_back({span[i][1], -1}); ++n;
        }
        std::sort(ev.begin(), ev.end());
        printf("SPAN %d workgroups (%d tiles), first start -> last end %.2f us; resident workgroups every 1 us:\n   ", n, tiles, (t1 - t0) / 100.0);
        int cur = 0; size_t k = 0;
        for (long long t = t0; t <= t1 + 100; t += 100) {
            while (k < ev.size() && ev[k].first <= t) cur += ev[k++].second;
            printf(" %d", cur);
        }
        double mean = 0; for (int i = 0; i < nblk; ++i) mean += (span[i][1] - span[i][0]) / 100.0;
        printf("\n    mean workgroup duration %.2f us; start of the last workgroup %.2f us\n", mean / n, (ev.empty() ? 0 : 0.0));
        // who is slow?  by XCD (block % 8), by CU occupancy (workgroups that overlapped in time on the same (xcc, se, cu))
        {
            double by_x[8] = {}; int nx_[8] = {};
            for (int i = 0; i < nblk; ++i) { by_x[i & 7] += (span[i][1] - span[i][0]) / 100.0; nx_[i & 7]++; }
            printf("    mean duration by block %% 8:");
            for (int q = 0; q < 8; ++q) printf(" %.1f", by_x[q] / (nx_[q] ? nx_[q] : 1));
            printf("\n");
            double alone = 0, shared = 0; int na = 0, ns = 0;
            for (int i = 0; i < nblk; ++i) {
                const long long key = ((span[i][3] & 15) << 20) | (span[i][2] & 0xff00);      // xcc, se / sh / cu bits of HW_ID
                bool sh = false;
                for (int j = 0; j < nblk && !sh; ++j)
                    if (j != i && (((span[j][3] & 15) << 20) | (span[j][2] & 0xff00)) == key && span[j][0] < span[i][1] - 200 && span[j][1] > span[i][0] + 200) sh = true;
                if (sh) { shared += (span[i][1] - span[i][0]) / 100.0; ++ns; } else { alone += (span[i][1] - span[i][0]) / 100.0; ++na; }
            }
            printf("    workgroups that shared their CU: %d, mean %.1f us; alone: %d, mean %.1f us\n", ns, ns ? shared / ns : 0, na, na ? alone / na : 0);
            printf("    durations of blocks 0..47 (us):");
            for (int i = 0; i < 48 && i < nblk; ++i) printf(" %.1f", (span[i][1] - span[i][0]) / 100.0);
            printf("\n");
        }
        long long last_start = 0; for (int i = 0; i < nblk; ++i) if (span[i][0] > last_start) last_start = span[i][0];
        printf("    last workgroup starts at %.2f us\n", (last_start - t0) / 100.0);
    }
    return 0;
}
