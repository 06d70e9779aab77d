#!/bin/bash
# Round 3, call 21: token_gemm (LDS-resident weight) with narrower weight blocks / more workgroups per CU against hipBLASLt.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03w; mkdir -p $O
cd $R
export TMPDIR=/tmp
for nb in 8 4 2; do MDETR_TOKEN_GEMM_NB=$nb python - <<PY
import torch
from monodetr_amd import token_gemm_ext
def t(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n * 1e3
out = []
for T, K, N in ((81600, 256, 256), (81600, 256, 384), (245760, 256, 64), (61440, 256, 512), (15360, 256, 1024), (15360, 256, 256)):
    x = torch.randn(T, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16) * 0.05; b = torch.randn(N, device="cuda").to(torch.bfloat16)
    lib = lambda: torch.nn.functional.linear(x, w, b)
    own = lambda: token_gemm_ext.token_gemm(x, w, b)
    err = float((lib().float() - own().float()).abs().max())
    out.append("%dx%d->%d lib %.1f own %.1f (err %.2g)" % (T, K, N, t(lib), t(own), err))
print("nb $nb:", " | ".join(out))
PY
done
