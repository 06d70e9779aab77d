#!/bin/bash
# rocprofv3 kernel trace of the bench command on the current tree -> per-category and per-kernel time per step
R=${GRAFT_REPO_ROOT:-$PWD}; T=${TAG:-r04t}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_step -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_traced.json 2>$O/bench_traced.err
cd $R; f=$(find /tmp/trace_step -name "*kernel_trace.csv" | head -1); st=$(find /tmp/trace_step -name "*kernel_stats.csv" | head -1)
python -m monodetr_amd.tools.trace_stats $f --steps 8 --out $O/${T}_bench_bf16_steady_kernel_stats.csv --top 200 > $O/${T}_trace_stats.txt 2>&1; head -16 $O/${T}_trace_stats.txt | cut -c1-170
grep -E "mdetr|Name" $st | head -120 > $O/${T}_rocprofv3_stats_mdetr_kernels.csv
tail -1 $O/bench_traced.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('traced', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
