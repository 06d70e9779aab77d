"""Op-level timing of the hand-written convolutions (csrc/conv3x3.hip, conv_taps.hip, conv_wgrad.hip, conv_stem.hip) against the
library convolutions they replace (MIOpen through aten), at the shapes of the training step: B = 8, 3 x 384 x 1280, bf16
channels_last.  TFLOP/s = 2 B OH OW K K C N / t; `frac_mfma` against the 2.5 PFLOP/s dense bf16 rate.

    python -m monodetr_amd.tools.convbench [--iters 20] [--only wgrad|strided|stem]
"""
import argparse
import json

import torch
import torch.nn.functional as F

from .fusedbench import timeit


def rec(ms, flops):
    return dict(ms=round(ms, 4), TFLOPs=round(flops / ms / 1e9, 1), frac_mfma=round(flops / (ms * 1e-3) / 2.5e15, 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev, B = "cuda", a.batch
    from monodetr_amd import conv3x3_ext, conv_taps_ext, conv_wgrad_ext
    conv_wgrad_ext.ENABLED = True
    res = {}

    def tensors(C, N, H, W, k, stride):
        pad = 1 if k == 3 else 0
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        x = torch.randn(B, C, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(N, C, k, k, device=dev) / (k * C ** 0.5)).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, N, OH, OW, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        return x, w, dy, pad, 2.0 * B * OH * OW * k * k * C * N

    # ---- forward / input gradient, 3x3 stride 1 (csrc/conv3x3.hip), kernel only: the PMC passes of scripts/r03_*.sh profile this ----
    if a.only == "conv3x3":
        for tag, (C, H, W) in (("layer1", (64, 96, 320)), ("layer2", (128, 48, 160)), ("layer3", (256, 24, 80)), ("layer4", (512, 12, 40))):
            x, w, dy, pad, flops = tensors(C, C, H, W, 3, 1)
            sh = torch.randn(C, device=dev)
            wo = conv3x3_ext._ohwi(w)
            res["conv3x3_%s_kernel" % tag] = rec(timeit(lambda: conv3x3_ext._launch(x, wo, sh, True), a.iters), flops)
            res["wgrad3x3_%s_kernel" % tag] = rec(timeit(lambda: conv_wgrad_ext.weight_gradient(x, dy, 3, 1, torch.bfloat16), a.iters), flops)

    # ---- weight gradient, 3x3 stride 1: the 10 trainable Bottleneck.conv2 of layer2-4 and the depth head's two convolutions ----
    if a.only in ("", "wgrad"):
        for tag, (C, H, W) in (("layer2", (128, 48, 160)), ("layer3", (256, 24, 80)), ("layer4", (512, 12, 40))):
            x, w, dy, pad, flops = tensors(C, C, H, W, 3, 1)
            res["wgrad3x3_%s_kernel" % tag] = rec(timeit(lambda: conv_wgrad_ext.weight_gradient(x, dy, 3, 1, torch.bfloat16), a.iters), flops)
            res["wgrad3x3_%s_library" % tag] = rec(timeit(lambda: torch.ops.aten.convolution_backward(
                dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False)), a.iters), flops)
            res["wgrad3x3_%s_kernel" % tag]["chunks"] = conv_wgrad_ext._lib().mdetr_conv_wgrad_chunks(B, H, W, C, H, W, C, 3, 1)

    # ---- stride-2 convolutions: forward, input gradient, weight gradient ----
    if a.only in ("", "strided"):
        shapes = (("layer2.0.conv2", 128, 128, 96, 320, 3), ("layer3.0.conv2", 256, 256, 48, 160, 3), ("layer4.0.conv2", 512, 512, 24, 80, 3),
                  ("input_proj.3", 2048, 256, 12, 40, 3), ("depth.downsample", 256, 256, 48, 160, 3),
                  ("layer2.0.downsample", 256, 512, 96, 320, 1), ("layer3.0.downsample", 512, 1024, 48, 160, 1), ("layer4.0.downsample", 1024, 2048, 24, 80, 1))
        for tag, C, N, H, W, k in shapes:
            x, w, dy, pad, flops = tensors(C, N, H, W, k, 2)
            sh = torch.randn(N, device=dev)
            shb = sh.to(torch.bfloat16)
            wo = conv_taps_ext._ohwi(w)
            res["fwd_%s_kernel" % tag] = rec(timeit(lambda: conv_taps_ext._forward(x, wo, sh, tag != "input_proj.3"), a.iters), flops)      # (the pyramid level has no ReLU: GroupNorm follows)
            res["fwd_%s_library" % tag] = rec(timeit(lambda: F.relu_(F.conv2d(x, w, shb, stride=2, padding=pad)), a.iters), flops)
            res["dgrad_%s_kernel" % tag] = rec(timeit(lambda: conv_taps_ext._input_gradient(dy, wo, H, W), a.iters), flops)
            res["dgrad_%s_library" % tag] = rec(timeit(lambda: torch.ops.aten.convolution_backward(
                dy, x, w, None, (2, 2), (pad, pad), (1, 1), False, (0, 0), 1, (True, False, False)), a.iters), flops)
            res["wgrad_%s_kernel" % tag] = rec(timeit(lambda: conv_wgrad_ext.weight_gradient(x, dy, k, 2, torch.bfloat16), a.iters), flops)
            res["wgrad_%s_library" % tag] = rec(timeit(lambda: torch.ops.aten.convolution_backward(
                dy, x, w, None, (2, 2), (pad, pad), (1, 1), False, (0, 0), 1, (False, True, False)), a.iters), flops)

    # ---- the stem: 7x7 / stride 2 / pad 3 on the 3-channel image (frozen: forward only) ----
    if a.only in ("", "stem"):
        try:
            from monodetr_amd import conv_stem_ext
            x = torch.randn(B, 3, 384, 1280, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            w = (torch.randn(64, 3, 7, 7, device=dev) / 12).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            sh = torch.randn(64, device=dev)
            shb = sh.to(torch.bfloat16)
            flops = 2.0 * B * 192 * 640 * 147 * 64
            packed = conv_stem_ext.pack_weight(w)
            res["stem_kernel"] = rec(timeit(lambda: conv_stem_ext._launch(x, packed, sh, True), a.iters), flops)
            res["stem_library"] = rec(timeit(lambda: F.relu_(F.conv2d(x, w, shb, stride=2, padding=3)), a.iters), flops)
        except ImportError:
            pass
    print(json.dumps(res))


if __name__ == "__main__":
    main()
