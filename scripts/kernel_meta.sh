#!/bin/bash
# developer tool: register / LDS / scratch figures of every kernel of one translation unit (device-only assembly listing)
#   scripts/kernel_meta.sh msda_fused.hip [extra flags]   -> /tmp/<name>.s + one line per kernel
R=$(cd $(dirname $0)/..; pwd); src=$1; shift
out=/tmp/$(basename $src .hip).s
cd $R/monodetr_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -Wno-pass-failed -I $R/include -I . "$@" --cuda-device-only -S $src -o $out
python3 - $out <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', txt, re.S):
    pass
for blk in txt.split('  - .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    print('%-90s vgpr %s agpr %s sgpr %s lds_static %s scratch %s spill_v %s' % (g('name')[:90], g('vgpr_count'), blk.split()[0], g('sgpr_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size'), g('vgpr_spill_count')))
PY
