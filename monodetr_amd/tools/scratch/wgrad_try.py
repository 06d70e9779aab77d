import torch, time
T, N, K = 81600, 256, 256
x = torch.randn(T, K, device="cuda", dtype=torch.bfloat16)
dy = torch.randn(T, N, device="cuda", dtype=torch.bfloat16)
def timeit(f, it=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
ref = (dy.float().t() @ x.float())
print("plain dy.t() @ x      : %.1f us" % timeit(lambda: dy.t() @ x))
for C in (32, 64, 96, 128, 160, 240, 480):
    if T % C: continue
    f = lambda C=C: torch.bmm(dy.view(C, T // C, N).transpose(1, 2), x.view(C, T // C, K)).sum(0)
    out = f()
    err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
    print("bmm split C=%3d        : %.1f us  rel err %.2e" % (C, timeit(f), err))
    f32 = lambda C=C: torch.bmm(dy.view(C, T // C, N).transpose(1, 2), x.view(C, T // C, K)).float().sum(0)
    print("   (+float before sum) : %.1f us" % timeit(f32))
print("bias: dy.sum(0)        : %.1f us" % timeit(lambda: dy.sum(0)))
print("bias via ones matmul   : %.1f us" % timeit(lambda: torch.ones(1, T, device='cuda', dtype=torch.bfloat16) @ dy))
