"""Op-level timing of the MSDA kernels with HIP events on the launch stream (developer tool).

    python -m monodetr_amd.tools.opbench [--hires] [--dist trained|uniform] [--iters 50]

Prints ms/call and achieved algorithmic GB/s (SURVEY.md section 8d formulas) for the encoder
(Lq = S) and decoder (Lq = 550 / 1100) shapes at B = 8.
"""
import argparse
import json

import torch

from monodetr_amd import msda_ext

KITTI = [(48, 160), (24, 80), (12, 40), (6, 20)]
KITTI_HI = [(64, 220), (32, 110), (16, 55), (8, 28)]


def algorithmic_bytes(B, S, M, D, L, Lq, P, e=4):
    fwd = e * B * (S * M * D + Lq * M * L * P * 3 + Lq * M * D)
    bwd = fwd + e * B * (S * M * D + Lq * M * L * P * 3)
    return fwd, bwd


def make(B, Lq, shapes, dist, seed=0, encoder=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    sh = torch.tensor(shapes, device="cuda")
    start = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    S = int(sh.prod(1).sum())
    M, D, L, P = 8, 32, len(shapes), 4
    value = torch.rand(B, S, M, D, device="cuda", generator=g) * 0.01
    if dist == "uniform":
        loc = torch.rand(B, Lq, M, L, P, 2, device="cuda", generator=g)
    else:   # "trained": reference point + N(0, 4 px) per level; "init": the module's
            # initial star pattern (head direction x point index 1..P px, ms_deform_attn.py:107-114)
        if encoder:
            refs = []
            for (H, W) in shapes:
                ys, xs = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, device="cuda") / H,
                                        torch.linspace(0.5, W - 0.5, W, device="cuda") / W, indexing="ij")
                refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
            ref = torch.cat(refs, 0)[None].expand(B, -1, -1)
        else:
            ref = torch.rand(B, Lq, 2, device="cuda", generator=g)
        wh = sh.flip(-1).float()
        if dist == "init":
            import math
            th = torch.arange(M, device="cuda", dtype=torch.float32) * (2.0 * math.pi / M)
            d = torch.stack([th.cos(), th.sin()], -1)
            d = d / d.abs().max(-1, keepdim=True)[0]
            off = d.view(1, 1, M, 1, 1, 2) * torch.arange(1, P + 1, device="cuda").view(1, 1, 1, 1, P, 1)
            off = off + 0.05 * torch.randn(B, Lq, M, L, P, 2, device="cuda", generator=g)
        elif dist.startswith("sigma"):
            off = torch.randn(B, Lq, M, L, P, 2, device="cuda", generator=g) * float(dist[5:])
        else:
            off = torch.randn(B, Lq, M, L, P, 2, device="cuda", generator=g) * 4.0
        loc = ref[:, :, None, None, None, :] + off / wh[None, None, None, :, None, :]
        loc = loc.contiguous()
    attn = torch.softmax(torch.randn(B, Lq, M, L * P, device="cuda", generator=g), -1).view(B, Lq, M, L, P)
    go = torch.randn(B, Lq, M * D, device="cuda", generator=g)
    return value, sh, start, loc, attn, go, (B, S, M, D, L, Lq, P)


def time_call(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(iters):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hires", action="store_true")
    ap.add_argument("--dist", default="trained")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"],
                    help="bf16 = the mixed-precision operator of the bf16 model body (bf16 value / out / grad_out)")
    a = ap.parse_args()
    from monodetr_amd import _capi
    names = {0: "msda_fwd", 1: "msda_bwd_d32", 2: "msda_scatter_tiles", 3: "msda_reduce_tiles", 6: "msda_bwd_fused", 7: "msda_absmax", 8: "msda_finalize"}
    shapes = KITTI_HI if a.hires else KITTI
    S = sum(h * w for h, w in shapes)
    res = {}
    for name, Lq, enc in (("encoder", S, True), ("decoder", 1100 if a.hires else 550, False)):
        v, sh, st, loc, attn, go, dims = make(a.B, Lq, shapes, a.dist, encoder=enc)
        fb, bb = algorithmic_bytes(*dims)
        if a.dtype == "bf16":
            v, go = v.to(torch.bfloat16), go.to(torch.bfloat16)
            e = 2
            fb = e * dims[0] * (dims[1] * 256 + Lq * 256) + 4 * dims[0] * Lq * 8 * 16 * 3
            bb = fb + 4 * dims[0] * (dims[1] * 256 + Lq * 8 * 16 * 3)
            f_fwd = lambda: msda_ext.ms_deform_attn_forward_bf16(v, sh, st, loc, attn)
            f_bwd = lambda: msda_ext.ms_deform_attn_backward_bf16(v, sh, st, loc, attn, go)
        else:
            f_fwd = lambda: msda_ext.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
            f_bwd = lambda: msda_ext.ms_deform_attn_backward(v, sh, st, loc, attn, go, 64)
        tf = time_call(f_fwd, a.iters)
        tb = time_call(f_bwd, a.iters)
        _capi.profile_enable(True)                            # per-kernel split of one more round, HIP events inside the C ABI
        for _ in range(10):
            f_bwd()
        torch.cuda.synchronize()
        _capi.profile_enable(False)
        split = {names.get(k, str(k)): round(ms / max(n, 1), 4) for k, key, n, ms in _capi.profile_read()}
        res[name] = dict(bwd_kernels_ms=split, Lq=Lq, fwd_ms=round(tf, 4), fwd_GBs=round(fb / tf / 1e6, 1), bwd_ms=round(tb, 4),
                         bwd_GBs=round(bb / tb / 1e6, 1), fwd_MB=round(fb / 1e6, 1), bwd_MB=round(bb / 1e6, 1))
    print(json.dumps(dict(dist=a.dist, hires=a.hires, B=a.B, **res)))


if __name__ == "__main__":
    main()
