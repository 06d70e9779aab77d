"""tgemm NN + threshold_backward against mdetr_tgemm_masked at the bottleneck shapes (hipGraph-replay timing)."""
import monodetr_amd._runtime_env  # noqa: F401
import torch
from monodetr_amd import tgemm_ext


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for name, T, K, N, res in (("l2_conv1_dgrad", 61440, 128, 512, True), ("l3_conv1_dgrad", 15360, 256, 1024, True), ("l4_conv1_dgrad", 3840, 512, 2048, True),
                           ("l2_conv3_dgrad", 61440, 512, 128, False), ("l3_conv3_dgrad", 15360, 1024, 256, False), ("l4_conv3_dgrad", 3840, 2048, 512, False)):
    gen = torch.Generator(device="cuda").manual_seed(1)
    dy = torch.randn(T, K, device="cuda", generator=gen).to(torch.bfloat16)
    w = (torch.randn(K, N, device="cuda", generator=gen) * 0.05).to(torch.bfloat16)
    x = torch.randn(T, N, device="cuda", generator=gen).clamp(min=0).to(torch.bfloat16)
    r = torch.randn(T, N, device="cuda", generator=gen).to(torch.bfloat16) if res else None
    sep = lambda: torch.ops.aten.threshold_backward(tgemm_ext.tgemm(dy, w, None, r, nn=True), x, 0.0)
    fused = lambda: tgemm_ext.tgemm_masked(dy, w, x, r)
    assert torch.equal(sep(), fused())
    only = timeit(lambda: tgemm_ext.tgemm(dy, w, None, r, nn=True))
    print("%-16s T=%6d K=%5d N=%5d  gemm %.1f  gemm+mask pass %.1f  masked gemm %.1f us" % (name, T, K, N, only, timeit(sep), timeit(fused)))
