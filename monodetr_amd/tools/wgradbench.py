"""Weight + bias gradient of the step's token-wise linear layers two ways in one process (developer tool):
the library route (batched split-K product + chunk sum + a two-launch column sum of dy) and csrc/conv_wgrad.hip's 1x1 case with the
bias gradient riding along (one kernel + one chunk sum).

    python -m monodetr_amd.tools.wgradbench [--iters 50] [--out gpurun_out/wgradbench.json]
"""
import argparse
import json

import monodetr_amd._runtime_env  # noqa: F401  (before torch)
import torch

SHAPES = [  # (name, T, K, N)
    ("encoder_256x256", 81600, 256, 256), ("encoder_packed_384", 81600, 256, 384), ("layer1_64to256", 245760, 64, 256),
    ("layer1_256to64", 245760, 256, 64), ("layer2_128to512", 61440, 128, 512), ("layer2_512to128", 61440, 512, 128),
    ("layer3_256to1024", 15360, 256, 1024), ("layer3_1024to256", 15360, 1024, 256), ("layer4_512to2048", 3840, 512, 2048),
    ("layer4_2048to512", 3840, 2048, 512), ("depth_tokens_256x256", 15360, 256, 256),
]


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
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from monodetr_amd import conv_wgrad_ext
    from monodetr_amd.monodetr import linear
    dev = torch.device("cuda", 0)
    res = {}
    for name, T, K, N in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(T + K + N)
        x = (torch.randn(T, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        dy = (torch.randn(T, N, generator=g) * 0.1).to(torch.bfloat16).to(dev)
        w = torch.zeros(N, K, dtype=torch.bfloat16, device=dev)
        row = {"T": T, "K": K, "N": N}
        for tag, on in (("library_ms", False), ("kernel_ms", True)):
            conv_wgrad_ext.TOKEN_ROUTE = on
            if on and not conv_wgrad_ext.token_supported(x, dy):
                row[tag] = None
                continue
            row[tag] = round(timeit(lambda: linear._weight_bias_grads(x, dy, w, True, True), a.iters), 4)
        res[name] = row
        print(name, row, flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
