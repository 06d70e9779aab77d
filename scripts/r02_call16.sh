#!/bin/bash
# Round 2, GPU call 16: small_wgrad kernel (tests, op timing against the library's products, the step with it on top of the
# committed list); graph / model tests with their full log.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02q; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_fused_gpu.py -x -q -p no:cacheprovider --timeout 500 -k "small_wgrad" > $O/pytest_sw.log 2>&1; grep -n "passed\|failed" $O/pytest_sw.log
timeout 600 python -X faulthandler -m pytest tests/test_graph_gpu.py tests/test_model_gpu.py -x -q -p no:cacheprovider --timeout 500 > $O/pytest_graph_model.log 2>&1; grep -n "passed\|failed" $O/pytest_graph_model.log
timeout 200 python - > $O/wgrad_timing.txt 2>&1 <<'PY'
import torch
from monodetr_amd import small_wgrad_ext, colsum_ext
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for T, N, K, dt in ((4400, 256, 256, torch.bfloat16), (4400, 512, 256, torch.bfloat16), (4400, 256, 256, torch.float32), (4400, 128, 256, torch.bfloat16), (8192, 256, 256, torch.bfloat16)):
    dy, x = torch.randn(T, N, device="cuda").to(dt), torch.randn(T, K, device="cuda").to(dt)
    C = 16
    lib = lambda: (dy.t() @ x, dy.sum(0))
    split = lambda: (colsum_ext.column_sum(torch.bmm(dy.view(C, T // C, N).transpose(1, 2), x.view(C, T // C, K)).view(C, -1), dt), colsum_ext.column_sum(dy, dt))
    own = lambda: small_wgrad_ext.small_wgrad(dy, x, dt)
    print("T=%d N=%d K=%d %s: library mm + sum %.1f us | bmm + 2 colsum %.1f us | small_wgrad %.1f us" % (T, N, K, str(dt)[6:], t(lib), t(split), t(own)))
PY
cat $O/wgrad_timing.txt
ALL="MDETR_FUSED_LOSSES=1 MDETR_FUSED_ADAMW=1 MDETR_FUSED_LN=1 MDETR_MSDA_PROLOGUE=1 MDETR_MSDA_BF16=1 MDETR_FUSED_EPILOGUE=1 MDETR_GEMM_RELU=1 MDETR_CONV3X3=1 MDETR_GROUP_NORM=1"
b() { timeout 400 env $1 python bench.py --no-cpu-baseline --no-variants 2>$O/bench_$2.err | tail -1 > $O/bench_$2.json; python -c "
import json; d=json.load(open('$O/bench_$2.json')); print('$2', {k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch'][:20], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"; }
b "X=1" committed
b "$ALL MDETR_SMALL_WGRAD=1" with_small_wgrad
