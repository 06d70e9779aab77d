"""The token-wise products of one training iteration (B = 8, 384 x 1280), each timed two ways in one process: the library GEMM
(+ its separate elementwise tail where the call site has one) and csrc/tgemm.hip with the tail in its epilogue (default tile, and
with --sweep every tile shape / pipeline depth).

Timing: `reps` launches captured into one hipGraph and replayed (no host launch gaps between kernels), operands rotating over enough
buffer sets to exceed the 256 MB Infinity Cache (a product never finds its own previous input there); microseconds per launch,
algorithmic bytes / time against 8 TB/s and flops / time against 2.5 PFLOP/s, the larger of the two floors as `bound_us`.

    python -m monodetr_amd.tools.gemmbench [--sweep] [--only enc,l3] [--reps 20] [--out gpurun_out/gemmbench.json]
"""
import argparse
import json
import os

import monodetr_amd._runtime_env  # noqa: F401  (before torch)
import torch
import torch.nn.functional as F

HBM, MFMA = 8.0e12, 2.5e15
# (name, T, K, N, tail) -- forward products y = x W^T; tail: "" | "relu" | "res_relu" | "relu_drop".  The input gradient of each
# (dX[T, K] = dY[T, N] W[N, K], NN form, "accum" where a residual-path gradient is summed in) is derived below.
T1, T2, T3, T4, TE, TD, TP = 245760, 61440, 15360, 3840, 81600, 4400, 15360
FORWARD = [
    ("l1_conv1_256to64", T1, 256, 64, "relu", False), ("l1_conv3_64to256", T1, 64, 256, "res_relu", False),
    ("l2b0_conv1_256to128", T1, 256, 128, "relu", True),
    ("l2_conv1_512to128", T2, 512, 128, "relu", True), ("l2_conv3_128to512", T2, 128, 512, "res_relu", True),
    ("l2_down_256to512", T2, 256, 512, "", True),
    ("l3b0_conv1_512to256", T2, 512, 256, "relu", True),
    ("l3_conv1_1024to256", T3, 1024, 256, "relu", True), ("l3_conv3_256to1024", T3, 256, 1024, "res_relu", True),
    ("l3_down_512to1024", T3, 512, 1024, "", True),
    ("l4b0_conv1_1024to512", T3, 1024, 512, "relu", True),
    ("l4_conv1_2048to512", T4, 2048, 512, "relu", True), ("l4_conv3_512to2048", T4, 512, 2048, "res_relu", True),
    ("l4_down_1024to2048", T4, 1024, 2048, "", True),
    ("proj0_512to256", T2, 512, 256, "", True), ("proj1_1024to256", T3, 1024, 256, "", True), ("proj2_2048to256", T4, 2048, 256, "", True),
    ("enc_256to256", TE, 256, 256, "", True), ("enc_ffn1_256to256", TE, 256, 256, "relu_drop", True), ("enc_packed_256to384", TE, 256, 384, "", True),
    ("dec_256to256", TD, 256, 256, "", True), ("dec_ffn1_256to256", TD, 256, 256, "relu_drop", True), ("dec_heads_256to1032", TD, 256, 1032, "", True),
    ("depth_256to256", TP, 256, 256, "", True),
]


EAGER = False                 # --eager (counter passes under rocprofv3): plain launches, HIP-event timed


def graph_time(fn_of_set, nsets, reps):
    """us per launch of fn_of_set(i) over `reps` launches replayed from one graph."""
    if EAGER:
        for i in range(nsets):
            fn_of_set(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(reps):
            fn_of_set(r % nsets)
        e1.record()
        e1.synchronize()
        return round(e0.elapsed_time(e1) * 1e3 / reps, 2)
    for i in range(nsets):
        fn_of_set(i)                                                # warm-up (lazy library initialisation outside the capture)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for r in range(reps):
                fn_of_set(r % nsets)
    best = None
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / reps
        best = t if best is None else min(best, t)
    del g
    return round(best, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sweep", action="store_true", help="every tile shape and pipeline depth of tgemm, not only the launcher's choice")
    ap.add_argument("--only", default="")
    ap.add_argument("--no-backward", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--tune", default="", help="extra tgemm variants as MDETR_TUNE strings, e.g. 'tgemm_tile=128x64,tgemm_pf=1;tgemm_grid=512'")
    ap.add_argument("--eager", action="store_true", help="plain launches instead of graph replays (counter passes)")
    a = ap.parse_args()
    global EAGER
    EAGER = a.eager
    from monodetr_amd import bias_act_ext, tgemm_ext
    dev = torch.device("cuda", 0)
    res = {}
    cases = []
    for name, T, K, N, tail, grad in FORWARD:
        cases.append((name, T, K, N, tail, False))
        if grad and not a.no_backward:
            cases.append((name + "_dgrad", T, N, K, "accum" if ("conv1" in name or name in ("enc_256to256", "enc_ffn1_256to256")) else "", True))
    for name, T, K, N, tail, nn in cases:
        if a.only and not any(k in name for k in a.only.split(",")):
            continue
        byts = 2 * (T * K + T * N * (2 if tail in ("res_relu", "accum") else 1) + N * K)
        flops = 2.0 * T * N * K
        nsets = max(1, min(6, int(300e6 // (2 * T * (K + 2 * N))) + 1))
        gen = torch.Generator(device="cpu").manual_seed(T + K + N)
        xs = [(torch.randn(T, K, generator=gen) * 0.5).to(torch.bfloat16).to(dev) for _ in range(nsets)]
        w = ((torch.randn(K, N, generator=gen) if nn else torch.randn(N, K, generator=gen)) * 0.05).to(torch.bfloat16).to(dev)
        b = torch.randn(N, generator=gen).to(torch.bfloat16).to(dev)
        rs = [torch.randn(T, N, generator=gen).to(torch.bfloat16).to(dev) for _ in range(nsets)] if tail in ("res_relu", "accum") else None
        ys = [torch.empty(T, N, dtype=torch.bfloat16, device=dev) for _ in range(nsets)]
        row = {"T": T, "K": K, "N": N, "tail": tail, "nn": nn, "bound_us": round(max(byts / HBM, flops / MFMA) * 1e6, 2),
               "hbm_us": round(byts / HBM * 1e6, 2), "mfma_us": round(flops / MFMA * 1e6, 2)}

        # ---- the library's way: GEMM (+ fused RELU_BIAS epilogue where it has one) + the separate tail the step runs today
        def lib(i):
            x = xs[i]
            if nn:
                if tail == "accum":
                    return rs[i].addmm_(x, w)
                return torch.mm(x, w, out=ys[i])
            if tail == "relu":
                return torch._addmm_activation(b, x, w.t(), out=ys[i])
            y = torch.addmm(b, x, w.t(), out=ys[i])
            if tail == "res_relu":
                return bias_act_ext.bias_act(y, None, rs[i], relu=True)
            if tail == "relu_drop":
                return bias_act_ext.bias_act(y, None, None, relu=True, dropout_p=0.1, seed=5)
            return y
        row["library_us"] = graph_time(lib, nsets, a.reps)

        # ---- tgemm, the tail inside
        def ours(i):
            x = xs[i]
            if nn:
                return tgemm_ext.tgemm(x, w, None, rs[i] if tail == "accum" else None, nn=True, out=rs[i] if tail == "accum" else ys[i])
            return tgemm_ext.tgemm(x, w, b, rs[i] if tail == "res_relu" else None, relu=tail in ("relu", "res_relu", "relu_drop"),
                                   out=ys[i], dropout_p=0.1 if tail == "relu_drop" else 0.0, seed=5)
        os.environ.pop("MDETR_TUNE", None)
        row["tgemm_us"] = graph_time(ours, nsets, a.reps)
        if a.sweep:
            for tile in ("128x128", "128x64", "64x128", "64x64"):
                for pf in ("1", "2"):
                    os.environ["MDETR_TUNE"] = "tgemm_tile=%s,tgemm_pf=%s" % (tile, pf)
                    row["tgemm_%s_pf%s_us" % (tile, pf)] = graph_time(ours, nsets, a.reps)
            os.environ.pop("MDETR_TUNE", None)
        for var in [v for v in a.tune.split(";") if v]:               # extra variants: "key=value,key=value;..." (each group timed once)
            os.environ["MDETR_TUNE"] = var
            row["tgemm[%s]_us" % var] = graph_time(ours, nsets, a.reps)
        os.environ.pop("MDETR_TUNE", None)
        best = min(v for k, v in row.items() if k.startswith("tgemm") and k.endswith("_us"))
        row["tgemm_best_us"], row["tgemm_frac_of_bound"] = best, round(row["bound_us"] / best, 3)
        # ---- agreement (one set): element-wise against the fp64 product is the tests' job; here the two paths side by side
        if tail != "accum" and tail != "relu_drop":
            y_l = lib(0).float().clone()
            y_o = ours(0).float()
            row["max_abs_diff_vs_library"] = float((y_l - y_o).abs().max())
        res[name] = row
        print(name, json.dumps(row), flush=True)
        del xs, ys, rs
        torch.cuda.empty_cache()
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
    tot_l = sum(r["library_us"] for r in res.values())
    tot_o = sum(r["tgemm_best_us"] for r in res.values())
    print("TOTAL library %.1f us, tgemm(best) %.1f us over %d products" % (tot_l, tot_o, len(res)))


if __name__ == "__main__":
    main()
