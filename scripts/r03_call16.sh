#!/bin/bash
# Round 3, call 16: conv_taps with 32-channel LDS slabs (alternate library) against 64, op level and step.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03q; mkdir -p $O
cd $R
export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', {k.replace('_kernel',''): v['ms'] for k, v in d.items() if k.endswith('_kernel') and ('$3' in k)})"; }
b() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'])" || tail -3 $O/bench_$tag.err; }
timeout 300 python -m monodetr_amd.tools.convbench --only strided --iters 20 2>/dev/null | tail -1 > $O/strided_slab64.json; show $O/strided_slab64.json slab64 fwd_; show $O/strided_slab64.json slab64 dgrad_
b slab64 X=1
cp monodetr_amd/libmonodetr_amd.so /tmp/lib64.so; cp monodetr_amd/libmonodetr_amd_slab32.so monodetr_amd/libmonodetr_amd.so
timeout 300 python -m monodetr_amd.tools.convbench --only strided --iters 20 2>/dev/null | tail -1 > $O/strided_slab32.json; show $O/strided_slab32.json slab32 fwd_; show $O/strided_slab32.json slab32 dgrad_
timeout 300 python -m pytest tests/test_fused_gpu.py -x -q -m gpu -p no:cacheprovider -k "conv" 2>&1 | tail -2
b slab32 X=1
cp /tmp/lib64.so monodetr_amd/libmonodetr_amd.so
