#!/bin/bash
# Round 3, call 17: conv_wgrad with 8 x 16-pixel tiles (half the LDS: two workgroups per CU) against 8 x 32, op level and step.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03s; mkdir -p $O
cd $R
export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', {k.replace('_kernel',''): v['ms'] for k, v in d.items() if k.endswith('_kernel') and ('$3' in k)})"; }
b() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'])" || tail -3 $O/bench_$tag.err; }
timeout 300 python -m monodetr_amd.tools.convbench --only wgrad --iters 20 2>/dev/null | tail -1 > $O/wgrad_cols32.json; show $O/wgrad_cols32.json cols32 wgrad
timeout 300 python -m monodetr_amd.tools.convbench --only strided --iters 20 2>/dev/null | tail -1 > $O/strided_cols32.json; show $O/strided_cols32.json cols32 wgrad_
b cols32 X=1
cp monodetr_amd/libmonodetr_amd.so /tmp/lib_main.so; cp monodetr_amd/libmonodetr_amd_alt.so monodetr_amd/libmonodetr_amd.so
for w in 256 512 768; do MDETR_CONV_WGRAD_WGS=$w timeout 300 python -m monodetr_amd.tools.convbench --only wgrad --iters 20 2>/dev/null | tail -1 > $O/wgrad_cols16_wgs$w.json; show $O/wgrad_cols16_wgs$w.json cols16_wgs$w wgrad; done
MDETR_CONV_WGRAD_WGS=512 timeout 300 python -m monodetr_amd.tools.convbench --only strided --iters 20 2>/dev/null | tail -1 > $O/strided_cols16.json; show $O/strided_cols16.json cols16_512 wgrad_
MDETR_CONV_WGRAD_WGS=512 timeout 300 python -m pytest tests/test_fused_gpu.py -x -q -m gpu -p no:cacheprovider -k "conv" 2>&1 | tail -2
b cols16_512 MDETR_CONV_WGRAD_WGS=512
cp /tmp/lib_main.so monodetr_amd/libmonodetr_amd.so
