#!/bin/bash
# Round 3, call 24: counters of the convolution kernels after the occupancy pass (conv3x3 with narrowed channel blocks, conv_wgrad with
# 8 x 16 tiles; the stride-2 kernels with 32-channel slabs), as r03b_pmc_conv.json before it.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03c2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_conv_$i -- python -m monodetr_amd.tools.convbench --only conv3x3 --iters 3 > $O/pmc_pass$i.log 2>&1 || echo "pmc pass $i failed"
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_taps_$i -- python -m monodetr_amd.tools.convbench --only strided --iters 2 > $O/pmc_taps_pass$i.log 2>&1 || echo "pmc taps pass $i failed"
done
cd $R
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_conv_* --match conv --out $O/r03_pmc_conv_after.json > /dev/null 2>$O/pmc_summary.err
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_taps_* --match conv --out $O/r03_pmc_conv_strided_after.json > /dev/null 2>>$O/pmc_summary.err
python -c "
import json
for f in ('r03_pmc_conv_after', 'r03_pmc_conv_strided_after'):
    for r in json.load(open('$O/' + f + '.json')):
        print(f[8:], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ('kernel', 'grid', 'workgroup', 'vgpr', 'lds_bytes', 'GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVES', 'frac_wave_parked', 'frac_issuing', 'lds_conflict_frac')})" | cut -c1-330 | head -40
