#!/bin/bash
# Library GEMM selection: the heuristic's choice against PyTorch's TunableOp timing the candidates during the eager start-up
# iterations (MDETR_BENCH_TUNABLEOP=1), same box
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04tun; mkdir -p $O
for v in 0 1; do
  s=$(date +%s)
  MDETR_BENCH_TUNABLEOP=$v timeout 130 python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-variants 2>$O/t$v.err | tail -1 > $O/t$v.json
  python -c "import sys,json,time; d=json.loads(open('$O/t$v.json').read()); print('TUNABLEOP=$v', d['value'], d['ms_per_step'], d['final_loss'], d['config'].get('gemm_selection'), 'wall', int(time.time())-$s, 's')" 2>&1 | tail -1
done
ls -la /tmp/mdetr_tunableop_results*.csv 2>/dev/null | head -3; wc -l /tmp/mdetr_tunableop_results*.csv 2>/dev/null | tail -1
cp /tmp/mdetr_tunableop_results*.csv $O/ 2>/dev/null
