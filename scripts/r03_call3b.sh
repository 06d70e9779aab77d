#!/bin/bash
# Round 3, call 3b (call 3 ran against a half-built library): graph_nan diagnostic, convolution tests, step A/B, and the first GPU
# run of the MSDA backward's owner scheme (MDETR_MSDA_OWNER=1): parity tests + operator timings against the candidate scheme.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
export TMPDIR=/tmp
for v in "" "--no-ref" "--sync-load" "--no-ref --lr 2e-4 --steps 16"; do
  echo "== graph_nan $v"; timeout 300 python tests/diag/graph_nan.py $v 2>&1 | grep -v Warning | grep "^i=\|twin\|Error\|error" | head -40
done > $O/graph_nan.log 2>&1
tail -70 $O/graph_nan.log | cut -c1-420
timeout 900 python -m pytest tests/test_fused_gpu.py -q -p no:cacheprovider --timeout 600 -k "conv_strided or conv_wgrad or conv_stem or convolution_kernels" > $O/pytest_conv.log 2>&1; echo "pytest conv rc=$?"
grep -n "passed\|failed\|^E  " $O/pytest_conv.log | cut -c1-1800 | tail -8
MDETR_MSDA_OWNER=1 timeout 900 python -m pytest tests/test_msda_gpu.py -q -p no:cacheprovider --timeout 600 > $O/pytest_msda_owner.log 2>&1; echo "pytest msda owner rc=$?"
grep -n "passed\|failed\|^E  " $O/pytest_msda_owner.log | cut -c1-600 | tail -8
for dist in init trained; do
  for ow in 0 1; do
    MDETR_MSDA_OWNER=$ow timeout 200 python -m monodetr_amd.tools.opbench --dtype bf16 --dist $dist --iters 30 > $O/opbench_${dist}_owner$ow.json 2>/dev/null
    python -c "import json; d=json.load(open('$O/opbench_${dist}_owner$ow.json')); print('$dist owner=$ow', d['encoder']['bwd_ms'], d['encoder']['bwd_kernels_ms'], 'decoder', d['decoder']['bwd_ms'])"
  done
done
for geo in "12 24 6" "20 28 2" "16 24 3" "24 32 0" "8 24 4"; do
  set -- $geo
  MDETR_MSDA_OWNER=1 MDETR_MSDA_OTILE_H=$1 MDETR_MSDA_OTILE_W=$2 MDETR_MSDA_OREACH=$3 timeout 200 python -m monodetr_amd.tools.opbench --dtype bf16 --dist init --iters 30 > $O/opbench_init_o$1x$2r$3.json 2>/dev/null
  python -c "import json; d=json.load(open('$O/opbench_init_o$1x$2r$3.json')); print('owner tile $1x$2 reach $3', d['encoder']['bwd_ms'], d['encoder']['bwd_kernels_ms'])"
done
MDETR_MSDA_OWNER=1 timeout 200 python -m monodetr_amd.tools.opbench --dist init --iters 30 > $O/opbench_init_fp32_owner1.json 2>/dev/null; python -c "import json; d=json.load(open('$O/opbench_init_fp32_owner1.json')); print('fp32 owner', d['encoder']['bwd_ms'], d['encoder']['bwd_kernels_ms'])"
ALL="MDETR_FUSED_LOSSES=1 MDETR_FUSED_ADAMW=1 MDETR_FUSED_LN=1 MDETR_MSDA_PROLOGUE=1 MDETR_MSDA_BF16=1 MDETR_FUSED_EPILOGUE=1 MDETR_GEMM_RELU=1 MDETR_CONV3X3=1 MDETR_GROUP_NORM=1 MDETR_SMALL_WGRAD=1"
for tag in committed wgrad all all_owner; do
  case $tag in committed) EXTRA="";; wgrad) EXTRA="MDETR_CONV_WGRAD=1";; all) EXTRA="MDETR_CONV_WGRAD=1 MDETR_CONV_STRIDED=1 MDETR_CONV_STEM=1";; all_owner) EXTRA="MDETR_CONV_WGRAD=1 MDETR_CONV_STRIDED=1 MDETR_CONV_STEM=1 MDETR_MSDA_OWNER=1";; esac
  env $ALL $EXTRA timeout 300 python bench.py --no-variants --no-cpu-baseline --steps 30 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['final_loss'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])" || tail -3 $O/bench_$tag.err
done
