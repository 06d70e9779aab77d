#!/bin/bash
# Round 3, call 10: any-N small_wgrad / split_rows MHA / depth-table gradient on the GPU; operator map with Python frames.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fused_gpu.py tests/test_attn_gpu.py -x -q -m gpu -p no:cacheprovider -k "small_wgrad or attn or step" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_subset.log
grep -n "^E  \|^FAILED" $O/pytest_subset.log | cut -c1-300 | head -12
timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'])" || tail -3 $O/bench.err
timeout 400 python -m monodetr_amd.tools.opmap --top 300 --out $O/opmap.txt --stacks aten::copy_,aten::cat,aten::fill_,aten::add,aten::add_,aten::sum,aten::mm,aten::threshold_backward,aten::clamp,aten::clamp_min,aten::div,aten::mul > /dev/null 2>$O/opmap.err; tail -2 $O/opmap.err
grep -c . $O/opmap.txt
