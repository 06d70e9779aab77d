#!/bin/bash
# Round 2, GPU call 7: the whole GPU suite with the final defaults, the committed bench line, and the two library-tuning
# experiments left over from round 1 (TunableOp for hipBLASLt, MIOpen find mode).
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02g; mkdir -p $O
cd $R
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 400 > $O/pytest_gpu_all.log 2>&1 ) 2>&1 | grep real
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest_gpu_all.log | cut -c1-300 | head -20
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %7.2f img/s %7.3f ms' % (sys.argv[1], d['value'], d['ms_per_step'])); r=d.get('roofline') or {}; print('    roofline frac', r.get('frac'), 'ms', r.get('avg_launch_ms'), r.get('kernel')); [print('   ', k, json.dumps(d[k])[:900]) for k in ('fp32_path','default_path','rccl_1rank','cpu_baseline') if k in d]" "$1"; }
( time timeout 500 python bench.py 2>$O/bench_err_committed.log | tee $O/bench_committed.json | val "committed (full line)" ) 2>&1 | grep -v "^$" | grep -v "^user\|^sys"
( time env MDETR_BENCH_TUNABLEOP=1 timeout 500 python bench.py --no-cpu-baseline --no-variants 2>$O/bench_err_tunable.log | tee $O/bench_tunableop.json | val "committed + TunableOp" ) 2>&1 | grep -v "^$" | grep -v "^user\|^sys"
( time env MDETR_BENCH_MIOPEN_FIND=1 timeout 500 python bench.py --no-cpu-baseline --no-variants 2>$O/bench_err_find.log | tee $O/bench_miopen_find.json | val "committed + MIOpen find" ) 2>&1 | grep -v "^$" | grep -v "^user\|^sys"
timeout 300 python bench.py --no-cpu-baseline --no-variants 2>/dev/null | tee $O/bench_committed_2.json | val "committed (again)"
