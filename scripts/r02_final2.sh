#!/bin/bash
# Round 2, last confirmation of the final tree: pytest -m gpu, smoke(), the default bench line, --config 5 (device matching for
# 100 queries per group -> graph replay), rocprofv3 --kernel-trace --stats of the bench command.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02v; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > $O/pytest_gpu_all.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" $O/pytest_gpu_all.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch']); print(d['roofline'])
for k in ('fp32_path','eager_path','default_path','rccl_1rank'): print(k, {a: b for a, b in d.get(k, {}).items() if a in ('value','ms_per_step','launch','error')})
print({k: v for k, v in d.get('cpu_baseline', {}).items() if k in ('value','cores','kind')})"
timeout 400 python bench.py --no-cpu-baseline --no-variants --config 5 2>$O/bench_config5.err | tail -1 > $O/bench_config5.json; python -c "
import json; d=json.load(open('$O/bench_config5.json')); print('config5', {k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch'][:60], d['roofline']['frac'])"
cd /tmp; PYTHONPATH=$R timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_step -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_traced.json 2>$O/bench_traced.err
cd $R; f=$(find /tmp/trace_step -name "*kernel_trace.csv" | head -1); st=$(find /tmp/trace_step -name "*kernel_stats.csv" | head -1)
python -m monodetr_amd.tools.trace_stats $f --steps 8 --out $O/r02v_bench_bf16_steady_kernel_stats.csv --top 6 > $O/trace_stats.txt 2>&1; head -30 $O/trace_stats.txt | cut -c1-160
grep -E "mdetr|Name" $st | head -60 > $O/r02v_rocprofv3_stats_mdetr_kernels.csv
tail -1 $O/bench_traced.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('traced', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
