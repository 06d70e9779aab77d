#!/bin/bash
# Round 3, call 22: attention with the key range split over two wave groups (550 x 1920): tests, op timing, step A/B.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03z; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_attn_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_attn.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_attn.log
MDETR_ATTN_KSPLIT=1 timeout 600 python -m pytest tests/test_attn_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_attn_forced.log 2>&1; echo "pytest (forced split) rc=$?"; tail -2 $O/pytest_attn_forced.log
for ks in 0 auto; do if [ $ks = auto ]; then unset MDETR_ATTN_KSPLIT; else export MDETR_ATTN_KSPLIT=$ks; fi
  timeout 300 python -m monodetr_amd.tools.attnbench --dropout 0.1 2>/dev/null | tail -1 > $O/attnbench_ks$ks.json; python -c "
import json; d=json.load(open('$O/attnbench_ks$ks.json')); print('ksplit $ks', {k: (v['hip']['fwd_TFLOPs'], v['hip']['bwd_TFLOPs'], v['hip']['fwd_ms'], v['hip']['bwd_ms']) for k, v in d.items() if isinstance(v, dict)})"; done
unset MDETR_ATTN_KSPLIT
b() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'])" || tail -3 $O/bench_$tag.err; }
b split X=1
b nosplit MDETR_ATTN_KSPLIT=0
