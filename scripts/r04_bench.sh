#!/bin/bash
# one bench line of the committed configuration (no side legs), kernel split of the MSDA operator inside the step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${TAG:-bench}; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants 2>$O/bench.err | tail -1 > $O/bench.json
python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('img/s', d['value'], 'ms', d['ms_per_step'], 'loss', d.get('final_loss'), 'roofline', d['roofline'])
for k in d.get('kernels', []):
    if 'msda' in k['kernel']: print('   ', k['kernel'], k.get('Lq'), k['avg_ms'], k.get('calls'))
PY
