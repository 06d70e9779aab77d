#!/bin/bash
# Round 6, evidence call: the driver's GPU commands on the current tree -- pytest -m gpu, smoke(), the default bench.py line --
# then the counter passes of the MSDA backward (the HBM traffic of the bench line's `roofline`; must precede bench.py: it checks
# the record's source hash), of tgemm / twgrad at the encoder shape and of the grouped fp32 head kernels (sgemm.hip), and the
# rocprofv3 kernel trace + stats of the bench command.
#   [SKIP_TESTS=1] [SKIP_PMC=1] bash scripts/r06_final.sh [tag]
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r06z}; O=$R/gpurun_out/$T; mkdir -p $O
cd $R
export TMPDIR=/tmp; rm -rf /tmp/pmc_msda_* /tmp/pmc_gemm_* /tmp/pmc_wgrad_* /tmp/trace_step      # (a reused box keeps /tmp: stale passes would be summarised with the new ones)
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider --timeout 600 > $O/pytest_gpu_all.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|error" $O/pytest_gpu_all.log | tail -3
grep -n "^E  \|^FAILED" $O/pytest_gpu_all.log | cut -c1-300 | head -12
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
if [ "${SKIP_PMC:-0}" != "1" ]; then
cd /tmp; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_msda_$i -- python -m monodetr_amd.tools.opbench --dist init --dtype bf16 --iters 3 > $O/pmc_msda_pass$i.log 2>&1 || echo "pmc msda pass $i failed" >> $O/errors.txt
done
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_gemm_$i -- python -m monodetr_amd.tools.gemmbench --eager --reps 6 --only enc_256to256,enc_ffn1_256to256,l2_conv3_128to512 > $O/pmc_gemm_pass$i.log 2>&1 || echo "pmc gemm pass $i failed" >> $O/errors.txt
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_wgrad_$i -- python -m monodetr_amd.tools.wgradbench --only encoder_256x256 --out /tmp/wg.json > $O/pmc_wgrad_pass$i.log 2>&1 || echo "pmc wgrad pass $i failed" >> $O/errors.txt
done
cd $R
bash scripts/r06_headspmc.sh ${T}_pmc_sgemm > $O/pmc_sgemm.txt 2>&1; cp $O/../${T}_pmc_sgemm/${T}_pmc_sgemm.json $O/ 2>/dev/null; tail -8 $O/pmc_sgemm.txt | cut -c1-260
bash scripts/r06_headstrace.sh > $O/${T}_heads_launch_times.txt 2>&1; cat $O/${T}_heads_launch_times.txt
bash scripts/r06_convpmc.sh ${T}_pmc_conv > $O/pmc_conv.txt 2>&1; cp $O/../${T}_pmc_conv/${T}_pmc_conv_pmc_conv.json $O/${T}_pmc_conv.json 2>/dev/null; tail -8 $O/pmc_conv.txt | cut -c1-200
python -m monodetr_amd.tools.convbench --only conv3x3 --iters 50 > $O/${T}_convbench_conv3x3.json 2>/dev/null; python -c "
import json; j=json.load(open('$O/${T}_convbench_conv3x3.json')); print('  '.join('%s %.1fus %.3f' % (k, v['ms']*1e3, v['frac_mfma']) for k, v in j.items()))"
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_msda_* --match msda --out $O/${T}_pmc_msda.json > /dev/null 2>$O/summary_msda.err
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_gemm_* --match tgemm --out $O/${T}_pmc_tgemm.json > /dev/null 2>$O/summary_gemm.err
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_wgrad_* --match twgrad --out $O/${T}_pmc_twgrad.json > /dev/null 2>$O/summary_wgrad.err
python -m monodetr_amd.tools.pmc_traffic_record $O/${T}_pmc_msda.json --out $O/msda_pmc_traffic.json --source profiles/${T}_pmc_msda.json | cut -c1-400
cp $O/msda_pmc_traffic.json profiles/msda_pmc_traffic.json
python - $O/${T}_pmc_tgemm.json $O/${T}_pmc_twgrad.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        rows = json.load(open(f))
    except Exception as e:
        print(f, e); continue
    for r in rows:
        print(r['kernel'][:70], r.get('grid'), 'cyc', round(r.get('GRBM_GUI_ACTIVE', 0) / 8), 'mfma', r.get('mfma_utilisation'), 'FETCH MB', round(r.get('FETCH_SIZE', 0) / 1024, 1), 'WRITE MB', round(r.get('WRITE_SIZE', 0) / 1024, 1))
PY
cat $O/errors.txt 2>/dev/null
fi
cd $R
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','final_loss','kernels_per_step','step_mfma_frac')}, d['config']['launch']); print(d['roofline'])
for k in ('fp32_path','eager_path','default_path','rccl_1rank','config2','config5'): print(k, {a: b for a, b in d.get(k, {}).items() if a in ('value','ms_per_step','error')})
for r in d.get('families', []): print(r['name'], r['launches'], r['ms_per_step'], r.get('frac'))
print({k: v for k, v in d.get('cpu_baseline', {}).items() if k in ('value','cores','kind','s_per_iter','note')})"
tail -3 $O/bench.err
cd /tmp; PYTHONPATH=$R MDETR_BENCH_FAMILIES=0 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_step -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_traced.json 2>$O/bench_traced.err
cd $R; f=$(find /tmp/trace_step -name "*kernel_trace.csv" | head -1); st=$(find /tmp/trace_step -name "*kernel_stats.csv" | head -1)
# (eight of the TIMED replays: the last steps of the process are eagerly launched side measurements; MDETR_BENCH_FAMILIES=0: no second profiler inside the traced process)
python -m monodetr_amd.tools.trace_stats $f --steps 8 --skip-last 14 --gaps 8 --out $O/${T}_bench_bf16_steady_kernel_stats.csv --top 14 > $O/trace_stats.txt 2>&1; grep -v " us/step " $O/trace_stats.txt | head -34 | cut -c1-170
grep -E "mdetr|Name" $st | head -90 > $O/${T}_rocprofv3_stats_mdetr_kernels.csv
tail -1 $O/bench_traced.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('traced', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline'].get('traffic'))"
python -m monodetr_amd.tools.rounds $f --steps 8 --top 60 > $O/${T}_workgroup_rounds.txt 2>&1
