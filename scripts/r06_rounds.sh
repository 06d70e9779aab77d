#!/bin/bash
# kernel trace of the replayed step -> workgroup rounds per launch (monodetr_amd/tools/rounds.py)
T=${1:-r06rounds}; R=$(pwd); O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp; D=/tmp/trace_rounds_$$_$RANDOM
cd /tmp; PYTHONPATH=$R MDETR_BENCH_FAMILIES=0 timeout 500 rocprofv3 --kernel-trace --output-format csv -d $D -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-variants > $O/bench_traced.json 2>$O/bench_traced.err
cd $R; f=$(find $D -name "*kernel_trace.csv" | head -1)
head -1 $f > $O/trace_header.txt
python -m monodetr_amd.tools.rounds $f --steps 4 --top 60 > $O/${T}_rounds.txt 2>$O/rounds.err; head -70 $O/${T}_rounds.txt | cut -c1-200; tail -3 $O/rounds.err
rm -rf $D
