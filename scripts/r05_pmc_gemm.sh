#!/bin/bash
# counters of tgemm and of the library's kernels on a few products of the step (separate --pmc passes, eager launches)
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-r05b}; O=$R/gpurun_out/$T; mkdir -p $O
ONLY=${ONLY:-enc_256to256,l3_conv1_1024to256,l2_conv1_512to128}
cd /tmp; export TMPDIR=/tmp; i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_gemm_$i -- python -m monodetr_amd.tools.gemmbench --eager --reps 6 --only $ONLY ${EXTRA:-} > $O/pmc_pass$i.log 2>&1 || echo "pmc pass $i failed" >> $O/errors.txt
done
cd $R
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_gemm_* --match tgemm --out $O/${T}_pmc_tgemm.json > /dev/null 2>$O/summary.err
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_gemm_* --match Cijk --out $O/${T}_pmc_library.json > /dev/null 2>>$O/summary.err
cat $O/errors.txt 2>/dev/null; tail -2 $O/pmc_pass1.log | cut -c1-300
python - $O/${T}_pmc_tgemm.json $O/${T}_pmc_library.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    for r in json.load(open(f)):
        print(r['kernel'][:64], r.get('grid'), 'v', r.get('vgpr'), 'lds', r.get('lds_bytes'), 'cyc', round(r.get('GRBM_GUI_ACTIVE', 0) / 8), 'mfma', r.get('mfma_utilisation'), 'ldsconf', r.get('lds_conflict_frac'),
              'issuing', r.get('frac_issuing'), 'parked', r.get('frac_wave_parked'), 'FETCH MB', round(r.get('FETCH_SIZE', 0) / 1024, 1), 'WRITE MB', round(r.get('WRITE_SIZE', 0) / 1024, 1), 'L2hit', r.get('L2_hit_rate'),
              'valu', round(r.get('SQ_INSTS_VALU', 0)), 'salu', round(r.get('SQ_INSTS_SALU', 0)), 'ldsi', round(r.get('SQ_INSTS_LDS', 0)), 'vmr', round(r.get('SQ_INSTS_VMEM_RD', 0)), 'waves', round(r.get('SQ_WAVES', 0)))
PY
