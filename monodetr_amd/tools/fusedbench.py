"""Op-level timing of the kernels written after round 1's GPU budget (the optional families of kernel_families.py) against the framework operators
they replace, at the encoder's token count (81 600 x 256, bf16), with algorithmic bytes -> fraction of the 8 TB/s HBM
roofline.

    python -m monodetr_amd.tools.fusedbench [--iters 30]
"""
import argparse
import json

import torch
import torch.nn.functional as F

HBM = 8.0e12


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


def row(ms, byts, **extra):
    return dict(ms=round(ms, 4), GBps=round(byts / ms / 1e6, 1), frac=round(byts / (ms * 1e-3) / HBM, 4), **extra)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    dev, T, C, e = "cuda", 81600, 256, 2
    res = {}

    # ---- residual + dropout + LayerNorm ----------------------------------------------------------------------------
    from monodetr_amd import add_ln_ext
    x = torch.randn(T, C, device=dev).to(torch.bfloat16).requires_grad_(True)
    r = torch.randn(T, C, device=dev).to(torch.bfloat16).requires_grad_(True)
    norm = torch.nn.LayerNorm(C).to(dev).to(torch.bfloat16)
    drop = torch.nn.Dropout(0.1)
    dy = torch.randn(T, C, device=dev).to(torch.bfloat16)
    fused_f = lambda: add_ln_ext.fused_add_layernorm(x, r, norm.weight, norm.bias, norm.eps, 0.1)
    torch_f = lambda: norm(x + drop(r))
    for tag, f in (("fused", fused_f), ("framework", torch_f)):
        y = f()
        bwd = lambda: torch.autograd.grad(y, (x, r, norm.weight, norm.bias), dy, retain_graph=True)
        res["add_ln_fwd_" + tag] = row(timeit(f, a.iters), 4 * T * C * e)
        res["add_ln_bwd_" + tag] = row(timeit(bwd, a.iters), 4 * T * C * e)

    # ---- token GEMM ------------------------------------------------------------------------------------------------
    from monodetr_amd import tgemm_ext
    w = (torch.randn(C, C, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(C, device=dev).to(torch.bfloat16)
    xt = x.detach()
    byts = e * (T * C + T * C + C * C)
    res["token_gemm_256x256"] = row(timeit(lambda: tgemm_ext.tgemm(xt, w, b), a.iters), byts, TFLOPs=None)
    res["library_gemm_256x256"] = row(timeit(lambda: F.linear(xt, w, b), a.iters), byts)
    res["token_gemm_256x256_relu"] = row(timeit(lambda: tgemm_ext.tgemm(xt, w, b, relu=True), a.iters), byts)
    res["library_gemm_256x256_relu"] = row(timeit(lambda: F.relu(F.linear(xt, w, b)), a.iters), byts)
    res["library_gemm_256x256_relu_epilogue"] = row(timeit(lambda: torch._addmm_activation(b, xt, w.t()), a.iters), byts)   # MDETR_GEMM_RELU

    # ---- convolution / FFN tails (csrc/bias_act.hip) -----------------------------------------------------------------
    from monodetr_amd import bias_act_ext
    for tag, shape in (("layer1_3x3_tail", (8, 64, 96, 320)), ("layer1_residual_tail", (8, 256, 96, 320)), ("layer3_residual_tail", (8, 1024, 24, 80))):
        act = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        skip = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if "residual" in tag else None
        shift = torch.randn(shape[1], device=dev).to(torch.bfloat16) if skip is None else None
        n = act.numel() * e
        res["bias_act_" + tag + "_fused"] = row(timeit(lambda: bias_act_ext.bias_act(act, shift, skip, relu=True), a.iters), n * (3 if skip is not None else 2))
        if skip is None:
            res["bias_act_" + tag + "_framework"] = row(timeit(lambda: F.relu_(act + shift.view(1, -1, 1, 1)), a.iters), n * 2)
        else:
            res["bias_act_" + tag + "_framework"] = row(timeit(lambda: F.relu_(act + skip), a.iters), n * 3)
    h = torch.randn(T, C, device=dev).to(torch.bfloat16).requires_grad_(True)
    for tag, f in (("fused", lambda: bias_act_ext.bias_act(h, None, None, relu=True, dropout_p=0.1)), ("framework", lambda: F.dropout(F.relu(h), 0.1, True))):
        y = f()
        res["ffn_relu_dropout_fwd_" + tag] = row(timeit(f, a.iters), 2 * T * C * e)
        res["ffn_relu_dropout_bwd_" + tag] = row(timeit(lambda: torch.autograd.grad(y, (h,), dy, retain_graph=True), a.iters), 3 * T * C * e)
    w128 = (torch.randn(128, C, device=dev) * 0.05).to(torch.bfloat16)
    byts = e * (T * C + T * 128 + 128 * C)
    res["token_gemm_256x128"] = row(timeit(lambda: tgemm_ext.tgemm(xt, w128), a.iters), byts)
    res["library_gemm_256x128"] = row(timeit(lambda: F.linear(xt, w128), a.iters), byts)

    # the backbone's layer1 expansions / reductions as token GEMMs (245 760 tokens)
    T1 = 8 * 96 * 320
    x64, x256 = torch.randn(T1, 64, device=dev).to(torch.bfloat16), torch.randn(T1, 256, device=dev).to(torch.bfloat16)
    w_up, w_dn = (torch.randn(256, 64, device=dev) * 0.1).to(torch.bfloat16), (torch.randn(64, 256, device=dev) * 0.05).to(torch.bfloat16)
    b_up, b_dn = torch.randn(256, device=dev).to(torch.bfloat16), torch.randn(64, device=dev).to(torch.bfloat16)
    byts = e * (T1 * 64 + T1 * 256 + 64 * 256)
    res["token_gemm_layer1_64to256"] = row(timeit(lambda: tgemm_ext.tgemm(x64, w_up, b_up), a.iters), byts)
    res["library_gemm_layer1_64to256"] = row(timeit(lambda: F.linear(x64, w_up, b_up), a.iters), byts)
    res["token_gemm_layer1_256to64_relu"] = row(timeit(lambda: tgemm_ext.tgemm(x256, w_dn, b_dn, relu=True), a.iters), byts)
    res["library_gemm_layer1_256to64_relu"] = row(timeit(lambda: F.relu_(F.linear(x256, w_dn, b_dn)), a.iters), byts)

    # ---- 3x3 convolutions of the backbone: csrc/conv3x3.hip (shift + ReLU inside) vs the library convolution + its tail ----
    from monodetr_amd import conv3x3_ext
    for tag, (Cc, Hc, Wc) in (("layer1", (64, 96, 320)), ("layer2", (128, 48, 160)), ("layer3", (256, 24, 80)), ("layer4", (512, 12, 40))):
        xc = torch.randn(8, Cc, Hc, Wc, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wc = (torch.randn(Cc, Cc, 3, 3, device=dev) / (3.0 * Cc ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        sh = torch.randn(Cc, device=dev)
        shb = sh.to(torch.bfloat16)
        flops = 18.0 * 8 * Hc * Wc * Cc * Cc
        byts = e * (2 * 8 * Hc * Wc * Cc + 9 * Cc * Cc)
        for name, f in (("conv3x3_" + tag + "_kernel", lambda: conv3x3_ext._launch(xc, conv3x3_ext._ohwi(wc), sh, True)),
                        ("conv3x3_" + tag + "_library", lambda: F.relu_(F.conv2d(xc, wc, shb, padding=1)))):
            ms = timeit(f, a.iters)
            res[name] = dict(row(ms, byts), TFLOPs=round(flops / ms / 1e9, 1), frac_mfma=round(flops / (ms * 1e-3) / 2.5e15, 4))

    # ---- fused AdamW -----------------------------------------------------------------------------------------------
    from monodetr_amd.helpers.optimizer_helper import AdamW, FusedAdamW
    for tag, cls in (("fused", FusedAdamW), ("foreach", AdamW)):
        params = [torch.nn.Parameter(torch.randn(256, 256, device=dev)) for _ in range(140)] + [torch.nn.Parameter(torch.randn(2048, 1024, device=dev)) for _ in range(12)]
        for p in params:
            p.grad = torch.randn_like(p)
        opt = cls([{'params': params, 'weight_decay': 1e-4}], lr=2e-4)
        opt.step()
        n = sum(p.numel() for p in params)
        res["adamw_" + tag] = row(timeit(opt.step, a.iters), 28 * n, parameters=n)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
