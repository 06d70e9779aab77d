#!/bin/bash
# Round 4, first GPU call for branch next/msda-prologue (none of it has run on a GPU): the MSDA tests, then the three kernels'
# times inside the step for (a) this tree, default lanes, (b) MDETR_MSDA_LPS=2, and -- when a build of main's msda_fused.hip is
# present as monodetr_amd/libmonodetr_amd_main.so (built here before the call:  git stash / checkout main -- csrc/msda_fused.hip,
# __graft_entry__.build(), cp, restore) -- (c) main's kernels in the same box.  One box, alternating runs: boxes differ by +-5 %.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_msda_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_msda.log 2>&1; echo "pytest default rc=$?"; tail -1 $O/pytest_msda.log
MDETR_MSDA_LPS=2 timeout 420 python -m pytest tests/test_msda_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_msda_lps2.log 2>&1; echo "pytest lps2 rc=$?"; tail -1 $O/pytest_msda_lps2.log
one() {   # name, env...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants 2>$O/bench_$name.err | tail -1 > $O/bench_$name.json
    python - "$O/bench_$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], 'img/s', d['value'], 'ms', d['ms_per_step'], 'loss', d['final_loss'], 'frac', d['roofline']['frac'], 'launch_ms', d['roofline']['avg_launch_ms'])
for k in d['kernels']:
    if 'msda' in k['kernel'] and k.get('Lq') == 10200 and 'backward' in k['kernel'] or 'absmax' in k['kernel'] or 'finalize' in k['kernel']:
        print('   ', k['kernel'], k.get('Lq'), k['avg_ms'])
PY
}
# the token GEMM forms against the library (one process, switch read per launch)
timeout 300 python -m monodetr_amd.tools.tokenbench --iters 50 --out $O/tokenbench.json > $O/tokenbench.log 2>&1; echo "tokenbench rc=$?"; cat $O/tokenbench.log | cut -c1-260
for rep in 1 2; do
    one cand_$rep MDETR_NOOP=1
    one lps2_$rep MDETR_MSDA_LPS=2
    if [ -f monodetr_amd/libmonodetr_amd_main.so ]; then
        cp monodetr_amd/libmonodetr_amd.so /tmp/cand.so; cp monodetr_amd/libmonodetr_amd_main.so monodetr_amd/libmonodetr_amd.so
        one main_$rep MDETR_NOOP=1
        cp /tmp/cand.so monodetr_amd/libmonodetr_amd.so
    fi
done
