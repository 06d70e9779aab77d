"""Token-shaped products y[T, N] = x[T, K] W^T + b (bf16) four ways in one process: the library GEMM, csrc/token_gemm.hip with the
inputs staged through LDS, its direct form (MDETR_TOKEN_GEMM_DIRECT=1: the switch is read at every launch) and the
weight-in-registers form (= 2) -- milliseconds,
algorithmic bytes / time as a fraction of the 8 TB/s HBM roofline, and whether the two kernel forms agree bit for bit.

    python -m monodetr_amd.tools.tokenbench [--iters 50] [--out gpurun_out/tokenbench.json]
"""
import argparse
import json
import os

import monodetr_amd._runtime_env  # noqa: F401  (before torch)
import torch
import torch.nn.functional as F

HBM = 8.0e12
SHAPES = [  # (name, T, K, N, relu)
    ("encoder_256to256", 81600, 256, 256, False),
    ("encoder_ffn_256to256_relu", 81600, 256, 256, True),
    ("encoder_256to128", 81600, 256, 128, False),
    ("encoder_packed_256to384", 81600, 256, 384, False),
    ("layer1_64to256", 245760, 64, 256, False),
    ("layer1_256to64_relu", 245760, 256, 64, True),
    ("layer1_64to64_relu", 245760, 64, 64, True),
    ("layer2_128to512", 61440, 128, 512, False),
    ("layer2_512to128_relu", 61440, 512, 128, True),
    ("decoder_256to256", 4400, 256, 256, False),
    ("decoder_256to256_relu", 4400, 256, 256, True),
    ("decoder_256to128", 4400, 256, 128, False),
    ("depth_tokens_256to256", 15360, 256, 256, False),
    ("layer3_256to1024", 15360, 256, 1024, False),
    ("layer3_512to256_relu", 15360, 512, 256, True),
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
    ap.add_argument("--ablate", action="store_true", help="time the weight-in-registers form with MDETR_TOKEN_GEMM_ABLATE = 1..7 "
                    "(bit 0 no stores, bit 1 inputs from L2, bit 2 no products)")
    ap.add_argument("--only", default="", help="comma-separated substrings: only the shapes whose name contains one of them")
    a = ap.parse_args()
    from monodetr_amd import token_gemm_ext
    dev = torch.device("cuda", 0)
    res = {}
    for name, T, K, N, relu in SHAPES:
        if a.only and not any(k in name for k in a.only.split(",")):
            continue
        g = torch.Generator(device="cpu").manual_seed(T + K + N)
        x = (torch.randn(T, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        b = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
        byts = 2 * (T * K + T * N + N * K)
        row = {"T": T, "K": K, "N": N, "relu": relu, "floor_ms": round(byts / HBM * 1e3, 4)}
        lib = (lambda: F.relu_(F.linear(x, w, b))) if relu else (lambda: F.linear(x, w, b))
        row["library_ms"] = round(timeit(lib, a.iters), 4)
        outs = {}
        for tag, flag in (("staged", "0"), ("direct", "1"), ("regs", "2")):      # (2: the weight-in-registers form, K <= 256)
            os.environ["MDETR_TOKEN_GEMM_DIRECT"] = flag
            if not token_gemm_ext.supported(x, w):
                continue
            f = lambda: token_gemm_ext.token_gemm(x, w, b, relu=relu)
            outs[tag] = f()
            row[tag + "_ms"] = round(timeit(f, a.iters), 4)
            row[tag + "_frac"] = round(byts / (row[tag + "_ms"] * 1e-3) / HBM, 4)
        if a.ablate and "regs_ms" in row:                          # developer timing of the weight-in-registers form with parts removed
            os.environ["MDETR_TOKEN_GEMM_DIRECT"] = "2"
            for bits in (1, 2, 3, 4, 5, 6, 7):
                os.environ["MDETR_TOKEN_GEMM_ABLATE"] = str(bits)
                row["regs_ablate%d_ms" % bits] = round(timeit(lambda: token_gemm_ext.token_gemm(x, w, b, relu=relu), a.iters), 4)
            os.environ.pop("MDETR_TOKEN_GEMM_ABLATE", None)
        if len(outs) >= 2:
            row["forms_bit_equal"] = all(bool(torch.equal(outs["staged"], o)) for o in outs.values())
        if outs:
            ref = lib().float()
            row["max_err_vs_library"] = float((next(iter(outs.values())).float() - ref).abs().max())
        res[name] = row
        print(name, row, flush=True)
    os.environ.pop("MDETR_TOKEN_GEMM_DIRECT", None)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
