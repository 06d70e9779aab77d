#!/bin/bash
# hardware counters of csrc/conv3x3.hip / conv_wgrad.hip at the four ResNet stages (B = 8): three passes, summary -> gpurun_out/<tag>/<tag>_pmc_conv.json
T=${1:-r06convpmc}; TUNE=${2:-}
R=$(pwd); O=$R/gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp; D=/tmp/pmc_conv_$$_$RANDOM; cd /tmp; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  MDETR_TUNE="$TUNE" PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d ${D}_$i -- python -m monodetr_amd.tools.convbench --only conv3x3 --iters 3 > $O/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python -m monodetr_amd.tools.pmc_summary ${D}_* --match conv --out $O/${T}_pmc_conv.json > /dev/null 2> $O/summary.err
rm -rf ${D}_*
python - $O/${T}_pmc_conv.json <<'P'
import json, sys
for r in json.load(open(sys.argv[1])):
    wc = r.get("SQ_WAVE_CYCLES") or 1
    g = lambda k: r.get(k) or 0
    print("%-46s grid %7d vgpr %3d lds %6d | mfma_util %s parked %s stall %s issuing %s | of wave cycles: wait_lds %.3f act_lds %.3f act_vmem %.3f act_valu %.3f misc %.3f | lds_idx/busy %.3f conflict %s fifo_full d %.3f c %.3f | lvl_lds/insts %.1f lvl_vmem/insts %.1f" % (
        r["kernel"][:46], r["grid"], r.get("vgpr", 0), r.get("lds_bytes", 0), r.get("mfma_utilisation"), r.get("frac_wave_parked"), r.get("frac_issue_stall"), r.get("frac_issuing"),
        g("SQ_WAIT_INST_LDS") / wc, g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_ACTIVE_INST_VMEM") / wc, g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_MISC") / wc,
        g("SQ_LDS_IDX_ACTIVE") / (g("SQ_BUSY_CYCLES") or 1), r.get("lds_conflict_frac"), g("SQ_LDS_DATA_FIFO_FULL") / wc, g("SQ_LDS_CMD_FIFO_FULL") / wc,
        g("SQ_INST_LEVEL_LDS") / (g("SQ_INSTS_LDS") or 1), g("SQ_INST_LEVEL_VMEM") / (g("SQ_INSTS_VMEM_RD") or 1)))
P
