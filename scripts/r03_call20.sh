#!/bin/bash
# Round 3, call 20: big-T token weight gradients through conv_wgrad<1,1> against the batched split-K product; step A/B.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03v; mkdir -p $O
cd $R
export TMPDIR=/tmp
MDETR_TOKEN_WGRAD_CONV=1 python - <<'PY'
import torch
from monodetr_amd import conv_wgrad_ext, colsum_ext
from monodetr_amd.monodetr.linear import _split_count
def t(f, n=30):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for T, K, N in ((81600, 256, 256), (81600, 256, 384), (245760, 64, 256), (245760, 256, 64), (61440, 512, 128), (61440, 128, 512), (61440, 256, 512), (15360, 1024, 256), (15360, 256, 1024), (15360, 512, 2048), (15360, 256, 256)):
    x = torch.randn(T, K, device="cuda").to(torch.bfloat16); dy = torch.randn(T, N, device="cuda").to(torch.bfloat16)
    C = _split_count(T)
    def lib():
        parts = torch.bmm(dy.view(C, T // C, -1).transpose(1, 2), x.view(C, T // C, -1))
        return colsum_ext.column_sum(parts.view(C, -1), torch.bfloat16)
    own = lambda: conv_wgrad_ext.token_weight_gradient(x, dy, torch.bfloat16)
    a, b = lib().view(N, K).float(), own().float()
    err = float((a - b).abs().max() / a.abs().max())
    print("T=%d K=%d N=%d: bmm(split %d)+colsum %.1f us | conv_wgrad<1,1> %.1f us | rel diff %.1e" % (T, K, N, C, t(lib), t(own), err))
PY
b() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'])" || tail -3 $O/bench_$tag.err; }
b base X=1
b tokconv MDETR_TOKEN_WGRAD_CONV=1
