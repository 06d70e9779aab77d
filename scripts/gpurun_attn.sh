#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/attn; mkdir -p $O
python -m pytest tests/test_attn_gpu.py -x -q 2>&1 | tail -1
for d in 0.0 0.1; do python -m monodetr_amd.tools.attnbench --dtype bf16 --dropout $d 2>/dev/null | tail -1; done
python -m monodetr_amd.tools.attnbench --dtype fp32 --dropout 0.1 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d /tmp/pmc_attn_a -- python -m monodetr_amd.tools.attnbench --dtype bf16 --dropout 0.1 --iters 3 > /dev/null 2>&1
timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_attn_b -- python -m monodetr_amd.tools.attnbench --dtype bf16 --dropout 0.1 --iters 3 > /dev/null 2>&1
cd $R
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_attn_a /tmp/pmc_attn_b --match attn --out $O/pmc_attn_$1.json > /dev/null
python - <<PY
import json
for r in json.load(open("$O/pmc_attn_$1.json")):
    print(r["kernel"][-40:], r["grid"], "conflict", r.get("lds_conflict_frac"), "parked", r.get("frac_wave_parked"), "issuing", r.get("frac_issuing"), "valu", round(r.get("SQ_INSTS_VALU",0)), "gui", round(r.get("GRBM_GUI_ACTIVE",0)))
PY
