#!/bin/bash
# Round 2, GPU call 5: the one-pass MSDA backward with all loads of a batch of sample groups in flight and the next
# step's candidates prefetched; launch variants; counters; a kernel-trace profile of the committed step.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02e; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_msda_gpu.py tests/test_model_gpu.py::test_bf16_body_outputs_and_gradients_vs_fp32 -q -rA -p no:cacheprovider --timeout 300 > $O/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | cut -c1-300 | head -20
ob() { echo "== $1 ${2:-init} ${3:-bf16}"; env $1 timeout 120 python -m monodetr_amd.tools.opbench --dist ${2:-init} --dtype ${3:-bf16} --iters 30 2>&1 | tail -1 | tee $O/opbench_$(echo "$1$2$3" | tr -c 'A-Za-z0-9\n' '_').json | cut -c1-330; }
ob "MDETR_MSDA_BWD=fused"
ob "MDETR_MSDA_TILE_H=16 MDETR_MSDA_TILE_W=24"
ob "MDETR_MSDA_TILE_H=12 MDETR_MSDA_TILE_W=32"
ob "MDETR_MSDA_GROUPS=8"
ob "MDETR_MSDA_GROUPS=8 MDETR_MSDA_TILE_H=24 MDETR_MSDA_TILE_W=40"
ob "MDETR_MSDA_THREADS=1024"
ob "MDETR_MSDA_TILE_H=16 MDETR_MSDA_TILE_W=24 MDETR_MSDA_CHUNKS=16"
ob "MDETR_MSDA_TILE_H=16 MDETR_MSDA_TILE_W=24 MDETR_MSDA_CHUNKS=16 MDETR_MSDA_WHOLE_LEVEL_CELLS=100"
ob "MDETR_MSDA_BWD=fused" trained
ob "MDETR_MSDA_TILE_H=16 MDETR_MSDA_TILE_W=24" trained
ob "MDETR_MSDA_BWD=fused" init fp32
cd /tmp; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  PYTHONPATH=$R MDETR_MSDA_TILE_H=16 MDETR_MSDA_TILE_W=24 timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_fused_$i -- python -m monodetr_amd.tools.opbench --dist init --dtype bf16 --iters 3 > $O/pmc_pass$i.log 2>&1 || echo "pmc pass $i failed" >> $O/errors.txt
done
cd $R
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_fused_* --match msda_bwd_fused --out $O/r02e_pmc_msda_fused_16x24.json 2>$O/summary.err | python -c "
import sys, json
for r in json.load(sys.stdin): print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})"
cat $O/errors.txt 2>/dev/null
# kernel trace of the committed training step (steady state) -> per-kernel / per-category statistics
cd /tmp; PYTHONPATH=$R timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_step -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-variants > $O/bench_traced.json 2>$O/bench_traced.err
cd $R; f=$(find /tmp/trace_step -name "*kernel_trace.csv" | head -1); echo "trace: $f"
python -m monodetr_amd.tools.trace_stats $f --steps 8 --out $O/r02e_bench_bf16_steady_kernel_stats.csv --top 45 2>&1 | tail -70
