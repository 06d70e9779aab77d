#!/bin/bash
# Where the replayed step idles: kernel + memory-copy trace of the plain step and of its one-rank RCCL form on one box,
# idle intervals grouped by the launches either side (tools/trace_stats --gaps).  LEGS="plain ddp sync" (packets = pre-recorded graph packets; sync = a device
# synchronisation after every replay: the host never runs ahead)
R=${GRAFT_REPO_ROOT:-$PWD}; T=${TAG:-r04gap}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for leg in ${LEGS:-plain ddp}; do
  rm -rf /tmp/trace_$leg
  unset MDETR_BENCH_FORCE_DDP MDETR_BENCH_STEP_SYNC DEBUG_CLR_GRAPH_PACKET_CAPTURE
  [ $leg = packets ] && export DEBUG_CLR_GRAPH_PACKET_CAPTURE=1     # the runtime's pre-recorded AQL packets (off by default here: _runtime_env.py)
  [ $leg = ddp ] && export MDETR_BENCH_FORCE_DDP=1
  [ $leg = sync ] && export MDETR_BENCH_STEP_SYNC=1
  PYTHONPATH=$R timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trace_$leg -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_$leg.json 2>$O/bench_$leg.err
  f=$(find /tmp/trace_$leg -name "*kernel_trace.csv" | head -1); c=$(find /tmp/trace_$leg -name "*memory_copy_trace.csv" | head -1)
  head -3 $c > $O/${T}_${leg}_copies_head.txt
  (cd $R; python -m monodetr_amd.tools.trace_stats $f ${c:+--copies $c} --steps 8 --skip-last 14 --gaps 8 --context 8 --top 12 > $O/${T}_$leg.txt 2>&1)
  tail -1 $O/bench_$leg.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$leg', d['value'], d['ms_per_step'], d['config'].get('launch'))"
  head -3 $O/${T}_$leg.txt | cut -c1-200
done
