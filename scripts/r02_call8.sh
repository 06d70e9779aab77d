#!/bin/bash
# Round 2, GPU call 8: packed FMAs + interleaved chunk blocks in the one-pass MSDA backward; counters (HBM traffic) of the
# final kernel; rocprofv3 --kernel-trace --stats of the committed bench command.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02h; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_msda_gpu.py -q -p no:cacheprovider --timeout 300 2>&1 | tail -2
ob() { echo "== $1 ${2:-init} ${3:-bf16}"; env $1 timeout 120 python -m monodetr_amd.tools.opbench --dist ${2:-init} --dtype ${3:-bf16} --iters 30 2>&1 | tail -1 | tee $O/opbench_$(echo "$1$2$3" | tr -c 'A-Za-z0-9\n' '_').json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e, c = d['encoder'], d['decoder']
print('   encoder fwd %.4f bwd %.4f ms %s | decoder bwd %.4f ms %s' % (e['fwd_ms'], e['bwd_ms'], e['bwd_kernels_ms'], c['bwd_ms'], c['bwd_kernels_ms']))"; }
ob "MDETR_MSDA_BWD=fused"
ob "MDETR_MSDA_CHUNKS=16"
ob "MDETR_MSDA_CHUNKS=8"
ob "MDETR_MSDA_BWD=fused" trained
ob "MDETR_MSDA_BWD=fused" init fp32
ob "MDETR_MSDA_BWD=tiled"
cd /tmp; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_fused_$i -- python -m monodetr_amd.tools.opbench --dist init --dtype bf16 --iters 3 > $O/pmc_pass$i.log 2>&1 || echo "pmc pass $i failed" >> $O/errors.txt
done
cd $R
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_fused_* --match msda --out $O/r02h_pmc_msda.json 2>$O/summary.err | python -c "
import sys, json
for r in json.load(sys.stdin): print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k in ('kernel','grid','workgroup','vgpr','FETCH_SIZE','WRITE_SIZE','L2_hit_rate','frac_wave_parked','frac_issuing','SQ_INSTS_VALU','SQ_INSTS_SALU','GRBM_GUI_ACTIVE','lds_conflict_frac')})"
cat $O/errors.txt 2>/dev/null
cd /tmp; PYTHONPATH=$R timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_step -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_traced.json 2>$O/bench_traced.err
cd $R; f=$(find /tmp/trace_step -name "*kernel_trace.csv" | head -1); st=$(find /tmp/trace_step -name "*kernel_stats.csv" | head -1); echo "trace: $f stats: $st"
python -m monodetr_amd.tools.trace_stats $f --steps 8 --out $O/r02h_bench_bf16_steady_kernel_stats.csv --top 12 2>&1 | tail -32
grep -E "mdetr|Name" $st | head -40 > $O/r02h_rocprofv3_stats_mdetr_kernels.csv; head -12 $O/r02h_rocprofv3_stats_mdetr_kernels.csv | cut -c1-220
tail -1 $O/bench_traced.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']); [print(k) for k in d['kernels'] if 'msda' in k['kernel']]"
