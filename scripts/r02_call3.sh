#!/bin/bash
# Round 2, GPU call 3: first GPU run of the one-pass MSDA backward (msda_fused.hip): parity suite, op-level A/B against
# round 1's gather + tile scatter + reduce, the bf16 parity test with its full report, and the committed bench line.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02c; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_msda_gpu.py tests/test_model_gpu.py::test_bf16_body_outputs_and_gradients_vs_fp32 "tests/test_fused_gpu.py::test_msda_bf16_kernels_match_the_fp32_kernels_on_rounded_inputs" tests/test_fused_gpu.py::test_training_step_with_bf16_msda_matches_default -q -rA -s -p no:cacheprovider --timeout 300 > $O/pytest.log 2>&1
grep -E "^(PASSED|FAILED|ERROR)|passed|failed|lowest gradient|max .bf16|total loss" $O/pytest.log | cut -c1-700 | head -80
for dist in init trained; do for path in fused tiled; do for dt in fp32 bf16; do
  echo "== $dist $path $dt"; MDETR_MSDA_BWD=$path timeout 120 python -m monodetr_amd.tools.opbench --dist $dist --dtype $dt --iters 30 2>&1 | tail -1 | tee $O/opbench_${dist}_${path}_${dt}.json | cut -c1-900
done; done; done
echo "== uniform fused fp32"; MDETR_MSDA_BWD=fused timeout 120 python -m monodetr_amd.tools.opbench --dist uniform --iters 10 2>&1 | tail -1 | tee $O/opbench_uniform_fused_fp32.json | cut -c1-600
for tile in "12 32 5" "16 32 4" "16 32 6" "24 40 5" "8 32 5"; do set -- $tile
  echo "== tile $1x$2 reach $3"; MDETR_MSDA_TILE_H=$1 MDETR_MSDA_TILE_W=$2 MDETR_MSDA_REACH=$3 timeout 120 python -m monodetr_amd.tools.opbench --dist init --dtype bf16 --iters 30 2>&1 | tail -1 | cut -c1-330
done
for ch in 4 16; do echo "== chunks $ch"; MDETR_MSDA_CHUNKS=$ch timeout 120 python -m monodetr_amd.tools.opbench --dist init --dtype bf16 --iters 30 2>&1 | tail -1 | cut -c1-330; done
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %7.2f img/s %7.3f ms' % (sys.argv[1], d['value'], d['ms_per_step'])); print('    roofline', d.get('roofline')); [print('   ', k, d[k].get('value'), d[k].get('ms_per_step', ''), d[k].get('note','')) for k in ('fp32_path','default_path','rccl_1rank','cpu_baseline') if k in d]" "$1"; }
( time timeout 500 python bench.py 2>$O/bench_err_committed.log | tee $O/bench_committed.json | val "committed (full line)" ) 2>&1 | grep -v "^$" | grep -v "^user\|^sys"
MDETR_MSDA_BWD=tiled timeout 300 python bench.py --no-cpu-baseline --no-variants 2>/dev/null | tee $O/bench_committed_tiled_bwd.json | val "committed, round-1 MSDA backward"
tail -3 $O/bench_err_committed.log | grep -v amdgpu.ids
