#!/bin/bash
# Round 2, GPU call 4: the reworked one-pass MSDA backward (table-driven candidate decode, two groups in flight), its
# launch variants, hardware counters for it, the new backbone / bf16 parity tests, and the committed bench line.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02d; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_msda_gpu.py tests/test_backbone_parity_gpu.py tests/test_model_gpu.py::test_bf16_body_outputs_and_gradients_vs_fp32 -q -rA -s -p no:cacheprovider --timeout 300 > $O/pytest.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed|gradient of the whole|parameter tensors below|max .bf16|total loss" $O/pytest.log | cut -c1-1800 | head -40
ob() { echo "== $1"; env $1 timeout 120 python -m monodetr_amd.tools.opbench --dist ${2:-init} --dtype ${3:-bf16} --iters 30 2>&1 | tail -1 | tee $O/opbench_$(echo "$1$2$3" | tr -c 'A-Za-z0-9\n' '_').json | cut -c1-420; }
ob "MDETR_MSDA_BWD=fused"
ob "MDETR_MSDA_PIPE=0"
ob "MDETR_MSDA_THREADS=1024"
ob "MDETR_MSDA_TILE_H=24 MDETR_MSDA_TILE_W=40"
ob "MDETR_MSDA_TILE_H=12 MDETR_MSDA_TILE_W=32"
ob "MDETR_MSDA_CHUNKS=16"
ob "MDETR_MSDA_TILE_H=24 MDETR_MSDA_TILE_W=40 MDETR_MSDA_CHUNKS=16"
ob "MDETR_MSDA_BWD=fused" trained
ob "MDETR_MSDA_BWD=fused" init fp32
# hardware counters of the default variant, one pass per counter set
cd /tmp; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_fused_$i -- python -m monodetr_amd.tools.opbench --dist init --dtype bf16 --iters 3 > $O/pmc_pass$i.log 2>&1 || echo "pmc pass $i failed" >> $O/errors.txt
done
cd $R
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_fused_* --out $O/r02d_pmc_msda_fused.json > /dev/null 2>$O/summary.err
python - <<'PY' 2>&1 | head -60
import json, os
rows = json.load(open(os.environ.get("O", "gpurun_out/r02d") + "/r02d_pmc_msda_fused.json")) if os.path.exists("gpurun_out/r02d/r02d_pmc_msda_fused.json") else []
for r in rows:
    if "fused" in r["kernel"] or "fwd_rec" in r["kernel"] or "absmax" in r["kernel"] or "finalize" in r["kernel"]:
        print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
PY
cat $O/errors.txt 2>/dev/null
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %7.2f img/s %7.3f ms' % (sys.argv[1], d['value'], d['ms_per_step'])); print('    roofline', d.get('roofline')); [print('   ', k, json.dumps(d[k])[:600]) for k in ('fp32_path','default_path','rccl_1rank','cpu_baseline') if k in d]" "$1"; }
( time timeout 500 python bench.py 2>$O/bench_err_committed.log | tee $O/bench_committed.json | val "committed (full line)" ) 2>&1 | grep -v "^$" | grep -v "^user\|^sys"
