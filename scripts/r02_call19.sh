#!/bin/bash
# Round 2, GPU call 19: why the graph capture fails for --config 2 / 5 (traceback).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02s; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline --no-variants --config 2 --steps 5 --warmup 2 --prime 2 2>$O/bench_config2.err | tail -1 > $O/bench_config2.json; grep -n -A16 "capture not used" $O/bench_config2.err | cut -c1-200
timeout 300 python bench.py --no-cpu-baseline --no-variants --config 5 --steps 5 --warmup 2 --prime 2 2>$O/bench_config5.err | tail -1 > $O/bench_config5.json; grep -n -A16 "capture not used" $O/bench_config5.err | cut -c1-200
