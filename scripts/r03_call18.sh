#!/bin/bash
# Round 3, call 18: conv_wgrad 8 x 16 tiles at ~256 workgroups against 8 x 32 (step A/B); workgroup target of small_wgrad.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03t; mkdir -p $O
cd $R
export TMPDIR=/tmp
b() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'])" || tail -3 $O/bench_$tag.err; }
for w in 1152 768 512 384 256; do MDETR_SMALL_WGRAD_WGS=$w python - <<PY
import torch, time
from monodetr_amd import small_wgrad_ext
def t(f, n=50):
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / n * 1e3
out = []
for T, N, K, dt in ((4400, 256, 256, torch.bfloat16), (4400, 384, 256, torch.bfloat16), (4400, 1032, 256, torch.float32), (4400, 6, 256, torch.float32), (15360, 61, 256, torch.bfloat16)):
    dy, x = torch.randn(T, N, device="cuda").to(dt), torch.randn(T, K, device="cuda").to(dt)
    out.append("%dx%dx%d %s %.1f" % (T, N, K, str(dt)[6:10], t(lambda: small_wgrad_ext.small_wgrad(dy, x, dt))))
print("small_wgrad wgs $w (us incl. chunk sum):", " | ".join(out))
PY
done
b cols16 X=1
b cols16_sw512 MDETR_SMALL_WGRAD_WGS=512
cp monodetr_amd/libmonodetr_amd.so /tmp/lib_main.so; cp monodetr_amd/libmonodetr_amd_alt.so monodetr_amd/libmonodetr_amd.so
b cols32 X=1
cp /tmp/lib_main.so monodetr_amd/libmonodetr_amd.so
timeout 300 python -m monodetr_amd.tools.convbench --only strided --iters 20 2>/dev/null | tail -1 > $O/strided_cols16.json; python -c "
import json; d=json.load(open('$O/strided_cols16.json')); print('cols16', {k.replace('_kernel',''): v['ms'] for k, v in d.items() if k.endswith('_kernel') and 'wgrad_' in k})"
b cols16_b X=1
