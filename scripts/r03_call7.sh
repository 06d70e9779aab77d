#!/bin/bash
# Round 3, call 7: the driver's round-end commands on the current tree (pytest -m gpu, smoke(), bench.py), then the rocprofv3
# kernel trace of the bench command (kernel inventory after the convolution families joined the committed list).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider --timeout 600 > $O/pytest_gpu_all.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" $O/pytest_gpu_all.log | tail -3
grep -n "^E  \|^FAILED" $O/pytest_gpu_all.log | cut -c1-300 | head -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch'], d['config'].get('gpu_clocks')); print(d['roofline'])
for k in ('fp32_path','eager_path','default_path','rccl_1rank','config2','config5'): print(k, {a: b for a, b in d.get(k, {}).items() if a in ('value','ms_per_step','launch','error')})
print({k: v for k, v in d.get('cpu_baseline', {}).items() if k in ('value','cores','kind','s_per_iter','note')})"
tail -3 $O/bench.err
cd /tmp; PYTHONPATH=$R timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_step -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_traced.json 2>$O/bench_traced.err
cd $R; f=$(find /tmp/trace_step -name "*kernel_trace.csv" | head -1); st=$(find /tmp/trace_step -name "*kernel_stats.csv" | head -1)
python -m monodetr_amd.tools.trace_stats $f --steps 8 --out $O/r03h_bench_bf16_steady_kernel_stats.csv --top 12 > $O/trace_stats.txt 2>&1; head -34 $O/trace_stats.txt | cut -c1-170
grep -E "mdetr|Name" $st | head -80 > $O/r03h_rocprofv3_stats_mdetr_kernels.csv
tail -1 $O/bench_traced.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('traced', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
