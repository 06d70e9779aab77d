#!/bin/bash
# counters of the convolution / weight-gradient kernels (MFMA utilisation, LDS conflicts, HBM bytes): separate --pmc passes over
# tools/convbench and tools/wgradbench, summarised by tools/pmc_summary
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r04g}; O=$R/gpurun_out/$T; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_conv_$i -- python -m monodetr_amd.tools.convbench --iters 3 > $O/pmc_conv_pass$i.log 2>&1 || echo "pmc conv pass $i failed" >> $O/errors.txt
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_wgrad_$i -- python -m monodetr_amd.tools.wgradbench --iters 3 > $O/pmc_wgrad_pass$i.log 2>&1 || echo "pmc wgrad pass $i failed" >> $O/errors.txt
done
cd $R
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_conv_* --match conv --out $O/${T}_pmc_conv.json > /dev/null 2>$O/summary_conv.err
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_wgrad_* --match conv_wgrad --out $O/${T}_pmc_token_wgrad.json > /dev/null 2>$O/summary_wgrad.err
cat $O/errors.txt 2>/dev/null
python - $O/${T}_pmc_conv.json $O/${T}_pmc_token_wgrad.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    for r in json.load(open(f)):
        print(r['kernel'][:70], r.get('grid'), 'mfma_util', r.get('mfma_utilisation'), 'lds_conflict', r.get('lds_conflict_frac'), 'issuing', r.get('frac_issuing'), 'FETCH KiB', round(r.get('FETCH_SIZE', 0)), 'WRITE KiB', round(r.get('WRITE_SIZE', 0)))
PY
