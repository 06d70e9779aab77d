#!/bin/bash
# Round 3, call 2: the fixed trainer tests, the new convolution kernels (GPU parity tests, op-level timings vs MIOpen, PMC of
# conv3x3 / conv_wgrad), the step with and without them.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_trainer_gpu.py "tests/test_fused_gpu.py::test_train_val_entry_point_end_to_end" -q -p no:cacheprovider --timeout 900 -s > $O/pytest_trainer.log 2>&1; echo "pytest trainer rc=$?"
grep -n "passed\|failed\|PG-CHILD\|spread\|worst" $O/pytest_trainer.log | tail -12
timeout 900 python -m pytest tests/test_fused_gpu.py -q -p no:cacheprovider --timeout 600 -k "conv_strided or conv_wgrad or conv_stem or convolution_kernels or conv3x3" > $O/pytest_conv.log 2>&1; echo "pytest conv rc=$?"
grep -n "passed\|failed\|Error\|assert " $O/pytest_conv.log | tail -20
timeout 600 python -m monodetr_amd.tools.convbench --iters 20 > $O/convbench.json 2>$O/convbench.err; echo "convbench rc=$?"; tail -2 $O/convbench.err
python - <<PY
import json
d = json.load(open("$O/convbench.json"))
for k in sorted(d):
    if k.endswith("_kernel"):
        lib = d.get(k[:-7] + "_library")
        print("%-34s %8.4f ms %7.1f TF/s" % (k[:-7], d[k]["ms"], d[k]["TFLOPs"]), ("| library %8.4f ms  x%.2f" % (lib["ms"], lib["ms"] / d[k]["ms"])) if lib else "")
PY
ALL="MDETR_FUSED_LOSSES=1 MDETR_FUSED_ADAMW=1 MDETR_FUSED_LN=1 MDETR_MSDA_PROLOGUE=1 MDETR_MSDA_BF16=1 MDETR_FUSED_EPILOGUE=1 MDETR_GEMM_RELU=1 MDETR_CONV3X3=1 MDETR_GROUP_NORM=1 MDETR_SMALL_WGRAD=1"
for tag in committed wgrad strided all; do
  case $tag in committed) EXTRA="";; wgrad) EXTRA="MDETR_CONV_WGRAD=1";; strided) EXTRA="MDETR_CONV_WGRAD=1 MDETR_CONV_STRIDED=1";; all) EXTRA="MDETR_CONV_WGRAD=1 MDETR_CONV_STRIDED=1 MDETR_CONV_STEM=1";; esac
  env $ALL $EXTRA timeout 300 python bench.py --no-variants --no-cpu-baseline --steps 30 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['final_loss'], d['config']['launch'][:20])" || tail -3 $O/bench_$tag.err
done
# PMC of the convolution kernels (separate passes, counters only)
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  PYTHONPATH=$R timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_conv_$i -- python -m monodetr_amd.tools.convbench --only conv3x3 --iters 3 > $O/pmc_pass$i.log 2>&1 || echo "pmc pass $i failed"
done
cd $R
python -m monodetr_amd.tools.pmc_summary /tmp/pmc_conv_* --match conv --out $O/pmc_conv.json > /dev/null 2>$O/pmc_summary.err; ls -la $O | head -30
