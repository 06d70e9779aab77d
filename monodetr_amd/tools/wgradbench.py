"""Weight + bias gradient of the step's token-wise layers in one process (developer tool): csrc/twgrad.hip (transposing LDS reads)
against the 1x1 case of csrc/conv_wgrad.hip (MDETR_TUNE="twgrad=0") and, with --library, the library's batched split-K route -- each
INCLUDING its chunk sum, timed as graph replays over rotating operand sets (tools/gemmbench.graph_time).

    python -m monodetr_amd.tools.wgradbench [--reps 20] [--tune "twgrad_wgs=512;..."] [--out gpurun_out/wgradbench.json]
"""
import argparse
import json
import os

import monodetr_amd._runtime_env  # noqa: F401  (before torch)
import torch

from monodetr_amd.tools.gemmbench import graph_time

HBM, MFMA = 8.0e12, 2.5e15
SHAPES = [  # (name, T, K, N)
    ("encoder_256x256", 81600, 256, 256), ("encoder_packed_384", 81600, 256, 384), ("layer2b0_256to128", 245760, 256, 128),
    ("layer2_128to512", 61440, 128, 512), ("layer2_512to128", 61440, 512, 128), ("layer2_down_256to512", 61440, 256, 512),
    ("layer3_256to1024", 15360, 256, 1024), ("layer3_1024to256", 15360, 1024, 256), ("layer3_down_512to1024", 15360, 512, 1024),
    ("layer4_512to2048", 3840, 512, 2048), ("layer4_2048to512", 3840, 2048, 512), ("layer4_down_1024to2048", 3840, 1024, 2048),
    ("proj0_512to256", 61440, 512, 256), ("proj2_2048to256", 3840, 2048, 256), ("depth_tokens_256x256", 15360, 256, 256),
    ("decoder_256x256", 4400, 256, 256), ("decoder_ffn_256x1024", 4400, 256, 1024),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--tune", default="", help="extra variants of the new kernel as MDETR_TUNE strings, e.g. 'twgrad_wgs=512;twgrad_wgs=1024'")
    ap.add_argument("--library", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from monodetr_amd import conv_wgrad_ext
    from monodetr_amd.monodetr import linear
    dev = torch.device("cuda", 0)
    res = {}
    for name, T, K, N in SHAPES:
        if a.only and not any(k in name for k in a.only.split(",")):
            continue
        nsets = max(1, min(6, int(300e6 // (2 * T * (K + N))) + 1))
        g = torch.Generator(device="cpu").manual_seed(T + K + N)
        xs = [(torch.randn(T, K, generator=g) * 0.5).to(torch.bfloat16).to(dev) for _ in range(nsets)]
        dys = [(torch.randn(T, N, generator=g) * 0.1).to(torch.bfloat16).to(dev) for _ in range(nsets)]
        w = torch.zeros(N, K, dtype=torch.bfloat16, device=dev)
        byts, flops = 2 * T * (K + N) + 2 * N * K, 2.0 * T * N * K
        row = {"T": T, "K": K, "N": N, "bound_us": round(max(byts / HBM, flops / MFMA) * 1e6, 2)}
        f = lambda i: linear._weight_bias_grads(xs[i], dys[i], w, True, True)
        conv_wgrad_ext.TOKEN_ROUTE = conv_wgrad_ext.ENABLED = True
        os.environ.pop("MDETR_TUNE", None)
        row["twgrad_us"] = graph_time(f, nsets, a.reps) if conv_wgrad_ext.token_supported(xs[0], dys[0]) else None
        for var in [v for v in a.tune.split(";") if v]:
            os.environ["MDETR_TUNE"] = var
            row["twgrad[%s]_us" % var] = graph_time(f, nsets, a.reps)
        os.environ["MDETR_TUNE"] = "twgrad=0"
        row["conv1x1_us"] = graph_time(f, nsets, a.reps) if conv_wgrad_ext.token_supported(xs[0], dys[0]) else None
        os.environ.pop("MDETR_TUNE", None)
        if a.library:
            conv_wgrad_ext.TOKEN_ROUTE = False
            row["library_us"] = graph_time(f, nsets, a.reps)
            conv_wgrad_ext.TOKEN_ROUTE = True
        if row["twgrad_us"]:
            row["frac_of_bound"] = round(row["bound_us"] / row["twgrad_us"], 3)
        res[name] = row
        print(name, json.dumps(row), flush=True)
        del xs, dys
        torch.cuda.empty_cache()
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
