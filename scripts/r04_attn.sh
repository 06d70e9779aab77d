#!/bin/bash
# attention kernels: the tree's library and every variants/lib_*.so, attnbench with dropout 0.1, alternating
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${TAG:-attn}; mkdir -p $O
cd $R; export TMPDIR=/tmp
one() { local name=$1; shift
  env "$@" timeout 200 python -m monodetr_amd.tools.attnbench --dropout 0.1 2>/dev/null | tail -1 > $O/$name.json
  python -c "
import json; d=json.load(open('$O/$name.json'))
print('$name', {k: (v.get('fwd_ms'), v.get('fwd_TFLOPs'), v.get('bwd_ms'), v.get('bwd_TFLOPs')) if isinstance(v, dict) else v for k, v in d.items()})" 2>&1 | cut -c1-400; }
for rep in 1 2; do
  one tree_$rep MDETR_NOOP=1
  for v in $(ls monodetr_amd/variants/ 2>/dev/null | sed 's/lib_//; s/.so//'); do one ${v}_$rep MDETR_LIB_PATH=$R/monodetr_amd/variants/lib_$v.so; done
done
if [ -n "$PYTEST" ]; then for v in $(ls monodetr_amd/variants/ 2>/dev/null | sed 's/lib_//; s/.so//'); do MDETR_LIB_PATH=$R/monodetr_amd/variants/lib_$v.so timeout 600 python -m pytest tests/test_attn_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1; done; fi
