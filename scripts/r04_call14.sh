R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04n; mkdir -p $O
cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_trainer_gpu.py -x -q -m gpu -p no:cacheprovider > $O/pytest_trainer.log 2>&1; echo "pytest trainer rc=$?"; tail -3 $O/pytest_trainer.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench.err | tail -1 > $O/bench.json
python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('img/s', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'])
print('mfma', d.get('mfma'))
for k in ('fp32_path','eager_path','default_path','rccl_1rank','config2','config5'): print(k, {a: b for a, b in d.get(k, {}).items() if a in ('value','ms_per_step','launch','error')})
PY
tail -3 $O/bench.err
