#!/bin/bash
# Round 3, call 13: projection shortcuts as gather + token GEMM (tests, step A/B), kernel trace of the current tree.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03n; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py tests/test_backbone_parity_gpu.py -x -q -m gpu -p no:cacheprovider --timeout 600 -k "decimate or conv or backbone or step" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_subset.log
grep -n "^E  \|^FAILED" $O/pytest_subset.log | cut -c1-300 | head -12
b() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'], d['roofline']['avg_launch_ms'])" || tail -3 $O/bench_$tag.err; }
b new X=1
b taps MDETR_CONV_S2_GEMM=0
b new2 X=1
cd /tmp; PYTHONPATH=$R timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_step -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_traced.json 2>$O/bench_traced.err
cd $R; f=$(find /tmp/trace_step -name "*kernel_trace.csv" | head -1)
python -m monodetr_amd.tools.trace_stats $f --steps 8 --out $O/r03n_bench_bf16_steady_kernel_stats.csv --top 25 > $O/trace_stats.txt 2>&1; head -45 $O/trace_stats.txt | cut -c1-170
