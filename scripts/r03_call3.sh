#!/bin/bash
# Round 3, call 3: where the non-finite value of the replayed iteration comes from (tests/diag/graph_nan.py), the convolution
# kernels after the rework (merged input-gradient launch, conflict-free weight-gradient staging), their timings, the step A/B.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
export TMPDIR=/tmp
for v in "" "--no-ref" "--sync-load" "--no-ref --lr 2e-4 --steps 16"; do
  echo "== graph_nan $v"; timeout 300 python tests/diag/graph_nan.py $v 2>&1 | grep -v Warning | grep "^i=\|twin\|Error\|error" | head -40
done > $O/graph_nan.log 2>&1
tail -60 $O/graph_nan.log
timeout 900 python -m pytest tests/test_fused_gpu.py -q -p no:cacheprovider --timeout 600 -k "conv_strided or conv_wgrad or conv_stem or convolution_kernels" > $O/pytest_conv.log 2>&1; echo "pytest conv rc=$?"
grep -n "passed\|failed\|^E  " $O/pytest_conv.log | cut -c1-1500 | tail -12
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
for tag in committed wgrad all; do
  case $tag in committed) EXTRA="";; wgrad) EXTRA="MDETR_CONV_WGRAD=1";; all) EXTRA="MDETR_CONV_WGRAD=1 MDETR_CONV_STRIDED=1 MDETR_CONV_STEM=1";; esac
  env $ALL $EXTRA timeout 300 python bench.py --no-variants --no-cpu-baseline --steps 30 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['final_loss'], d['config']['launch'][:20])" || tail -3 $O/bench_$tag.err
done
