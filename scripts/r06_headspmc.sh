#!/bin/bash
# PMC passes over the grouped head kernels (tests/test_sgemm_gpu.py -k heads_level), summarised per (kernel, grid).
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r06_pmc_sgemm}; O=$R/gpurun_out/$T; mkdir -p $O; export TMPDIR=/tmp; cd /tmp; D=/tmp/pmc_$$_$RANDOM; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d ${D}_$i -- python -m pytest $R/tests/test_sgemm_gpu.py -q -m gpu -p no:cacheprovider -k heads_level > $O/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $R; python -m monodetr_amd.tools.pmc_summary ${D}_* --match sgemm --last 1 --out $O/$T.json > /dev/null
python - $O/$T.json <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    cyc = r.get("GRBM_GUI_ACTIVE", 0) / 8
    print(r["kernel"][-22:], r["grid"], "cycles %d" % cyc, "waves %d" % r.get("SQ_WAVES", 0), "mfma_util", r.get("mfma_utilisation"), "parked", r.get("frac_wave_parked"), "issue_stall", r.get("frac_issue_stall"), "issuing", r.get("frac_issuing"),
          "VALU %d SALU %d SMEM %d LDS %d VMRD %d VMWR %d FLAT %d" % tuple(r.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT")), "lds_conf", r.get("lds_conflict_frac"))
PY
rm -rf ${D}_*
