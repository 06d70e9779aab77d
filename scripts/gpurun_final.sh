#!/bin/bash
# end-of-round evidence: GPU tests, smoke, bench lines, steady-state kernel stats + rocprofv3 --stats, op benches
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/final; mkdir -p $O; T=${1:-r01h}
python -m pytest tests -m gpu -q 2>&1 | tail -2 > $O/${T}_pytest_gpu.log; cat $O/${T}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/${T}_bench_bf16.json 2> $O/bench_bf16.err; cut -c1-200 $O/${T}_bench_bf16.json
python bench.py --precision fp32 --no-cpu-baseline > $O/${T}_bench_fp32.json 2>/dev/null; cut -c1-200 $O/${T}_bench_fp32.json
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o t -- python $R/bench.py --steps 5 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); st=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
cd $R
python -m monodetr_amd.tools.trace_stats $f --steps 4 --out $O/${T}_bench_bf16_steady_kernel_stats.csv --top 3 2>&1 | grep window
[ -n "$st" ] && grep -E "Name|mdetr" $st > $O/${T}_rocprofv3_stats_mdetr_kernels.csv
python -m monodetr_amd.tools.opbench --dist trained 2>/dev/null | tail -1 > $O/${T}_opbench_trained.json; cat $O/${T}_opbench_trained.json | cut -c1-300
python -m monodetr_amd.tools.attnbench --dtype bf16 --dropout 0.1 2>/dev/null | tail -1 > $O/${T}_attnbench_bf16_p01.json
python -m monodetr_amd.tools.attnbench --dtype fp32 --dropout 0.1 2>/dev/null | tail -1 > $O/${T}_attnbench_fp32_p01.json
python -m monodetr_amd.tools.stepbreakdown --precision bf16 2>/dev/null | tail -2
