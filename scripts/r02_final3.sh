#!/bin/bash
# Round 2, confirmation of the last changes (pre-pass grid, per-stream workspaces): MSDA / colsum / fused / lsa tests, smoke, bench.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02x; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_colsum_gpu.py tests/test_fused_gpu.py tests/test_lsa_gpu.py -q -p no:cacheprovider --timeout 600 > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest_subset.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 python -m monodetr_amd.tools.opbench --dist init --dtype bf16 --iters 30 2>&1 | tail -1 > $O/opbench_final.json; python -c "
import json; d=json.load(open('$O/opbench_final.json')); e, c = d['encoder'], d['decoder']
print('encoder fwd %.4f bwd %.4f ms %s | decoder bwd %.4f ms %s' % (e['fwd_ms'], e['bwd_ms'], e['bwd_kernels_ms'], c['bwd_ms'], c['bwd_kernels_ms']))"
timeout 400 python bench.py --no-cpu-baseline --no-variants 2>$O/bench.err | tail -1 > $O/bench.json; python -c "
import json; d=json.load(open('$O/bench.json')); print({k: d[k] for k in ('value','ms_per_step','final_loss')}, d['config']['launch'][:20], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
