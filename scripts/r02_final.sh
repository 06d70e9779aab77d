#!/bin/bash
# Round 2, final evidence call: the driver's GPU commands on the final tree -- pytest -m gpu, smoke(), the default bench.py
# line -- then the rocprofv3 kernel trace + stats of the bench command, the counters (HBM traffic) of the final MSDA backward,
# and bench.py --config 2.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02t; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 > $O/pytest_gpu_all.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" $O/pytest_gpu_all.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch']); print(d['roofline'])
for k in ('fp32_path','eager_path','default_path','rccl_1rank'): print(k, {a: b for a, b in d.get(k, {}).items() if a in ('value','ms_per_step','launch','error')})
print({k: v for k, v in d.get('cpu_baseline', {}).items() if k in ('value','cores','kind')})"
cd /tmp; PYTHONPATH=$R timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_step -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_traced.json 2>$O/bench_traced.err
cd $R; f=$(find /tmp/trace_step -name "*kernel_trace.csv" | head -1); st=$(find /tmp/trace_step -name "*kernel_stats.csv" | head -1); echo "trace: $f stats: $st"
python -m monodetr_amd.tools.trace_stats $f --steps 8 --out $O/r02t_bench_bf16_steady_kernel_stats.csv --top 14 2>&1 | tail -24 | cut -c1-200
grep -E "mdetr|Name" $st | head -60 > $O/r02t_rocprofv3_stats_mdetr_kernels.csv; head -8 $O/r02t_rocprofv3_stats_mdetr_kernels.csv | cut -c1-200
tail -1 $O/bench_traced.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('traced', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
cd /tmp; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_fused_$i -- python -m monodetr_amd.tools.opbench --dist init --dtype bf16 --iters 3 > $O/pmc_pass$i.log 2>&1 || echo "pmc pass $i failed" >> $O/errors.txt
done
cd $R
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_fused_* --match msda --out $O/r02t_pmc_msda.json 2>$O/summary.err | python -c "
import sys, json
for r in json.load(sys.stdin): print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ('kernel','grid','workgroup','vgpr','FETCH_SIZE','WRITE_SIZE','L2_hit_rate','frac_wave_parked','frac_issuing','SQ_INSTS_VALU','SQ_INSTS_SALU','GRBM_GUI_ACTIVE','lds_conflict_frac')})"
cat $O/errors.txt 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-variants --config 2 2>$O/bench_config2.err | tail -1 > $O/bench_config2.json; python -c "
import json; d=json.load(open('$O/bench_config2.json')); print('config2', {k: d[k] for k in ('value','ms_per_step','dtype')}, d['config']['launch'][:40])"
