#!/bin/bash
# one PMC pass per counter set over the MSDA and attention op benches (run through gpurun)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ|TCC|TCP|TA|GRBM)_[A-Za-z0-9_]+" | sort -u > $O/counter_names.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_msda_$i -- python -m monodetr_amd.tools.opbench --dist trained --iters 3 > $O/msda_pass$i.log 2>&1 || echo "msda pass $i failed" >> $O/errors.txt
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_attn_$i -- python -m monodetr_amd.tools.attnbench --dtype bf16 --iters 3 > $O/attn_pass$i.log 2>&1 || echo "attn pass $i failed" >> $O/errors.txt
done
cd $R
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_msda_* --out $O/r01h_pmc_msda.json > /dev/null 2>$O/summary_msda.err
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_attn_* --match attn --out $O/r01h_pmc_attn.json > /dev/null 2>$O/summary_attn.err
ls -la $O; head -c 1500 $O/r01h_pmc_msda.json; cat $O/errors.txt 2>/dev/null; tail -3 $O/msda_pass1.log
