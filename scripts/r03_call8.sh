#!/bin/bash
# Round 3, call 8: validation of the packed MSDA projection, the fp32-value decoder cross-attention, the counted AdamW step and the
# mirrored conv3x3 input gradient; step A/B of each; operator map of the framework launches; kernel count.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03i; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_trainer_gpu.py tests/test_fused_gpu.py tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider --timeout 600 \
  -k "trainer or replay or process_group or adamw or conv3x3 or prologue or bf16_body or conv_step or e2e or msda" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_subset.log
grep -n "^E  \|^FAILED" $O/pytest_subset.log | cut -c1-300 | head -12
b() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'], d['roofline']['avg_launch_ms'])" || tail -3 $O/bench_$tag.err; }
b new X=1
b unpacked MDETR_MSDA_PACKED=0
b narrow MDETR_MSDA_WIDE_VALUE=0
timeout 400 python -m monodetr_amd.tools.opmap --top 400 --out $O/opmap.txt > /dev/null 2>$O/opmap.err; head -60 $O/opmap.txt | cut -c1-200; tail -2 $O/opmap.err
cd /tmp; PYTHONPATH=$R timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_step -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_traced.json 2>$O/bench_traced.err
cd $R; f=$(find /tmp/trace_step -name "*kernel_trace.csv" | head -1)
python -m monodetr_amd.tools.trace_stats $f --steps 8 --out $O/r03i_bench_bf16_steady_kernel_stats.csv --top 12 > $O/trace_stats.txt 2>&1; head -30 $O/trace_stats.txt | cut -c1-170
