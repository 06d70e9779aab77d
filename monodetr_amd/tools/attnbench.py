"""Op-level timing of the fused attention kernels vs PyTorch SDPA at the three hot-path shapes (B = 8).

    python -m monodetr_amd.tools.attnbench [--dtype bf16|fp32] [--iters 30]
"""
import argparse
import json

import torch
import torch.nn.functional as F

from monodetr_amd.attn_ext import fused_attention

SHAPES = [("depth_encoder_self", 8, 1920, 1920), ("depth_cross", 8, 550, 1920), ("grouped_self", 88, 50, 50)]


def timeit(fn, iters):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dropout", type=float, default=0.0, help="attention dropout probability (the model trains with 0.1)")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    H, E = 8, 256
    res = {}
    for name, B, Lq, Lk in SHAPES:
        q, k, v = (torch.randn(B, L, E, device="cuda", dtype=dt, requires_grad=True) for L in (Lq, Lk, Lk))
        go = torch.randn(B, Lq, E, device="cuda", dtype=dt)
        flops = 4.0 * B * H * Lq * Lk * 32

        def hip_f():
            return fused_attention(q, k, v, H, dropout_p=a.dropout)

        def sdpa_f():
            qh, kh, vh = (t.view(B, -1, H, 32).transpose(1, 2) for t in (q, k, v))
            return F.scaled_dot_product_attention(qh, kh, vh, dropout_p=a.dropout).transpose(1, 2).reshape(B, Lq, E)

        row = {}
        for tag, f in (("hip", hip_f), ("sdpa", sdpa_f)):
            tf = timeit(lambda: f(), a.iters)
            out = f()
            tb = timeit(lambda: torch.autograd.grad(out, (q, k, v), go, retain_graph=True), a.iters)
            row[tag] = dict(fwd_ms=round(tf, 4), bwd_ms=round(tb, 4), fwd_TFLOPs=round(flops / tf / 1e9, 1),
                            bwd_TFLOPs=round(2.5 * flops / tb / 1e9, 1))
        res[name] = row
    print(json.dumps(dict(dtype=a.dtype, dropout=a.dropout, **res)))


if __name__ == "__main__":
    main()
