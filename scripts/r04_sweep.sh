#!/bin/bash
# knob sweeps on the final kernels (one box): token weight-gradient workgroup target, MSDA backward tile / chunk counts
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${TAG:-sweep}; mkdir -p $O
cd $R; export TMPDIR=/tmp
for w in 256 192 384 512 768; do
  echo "== MDETR_CONV_WGRAD_WGS=$w"; MDETR_CONV_WGRAD_WGS=$w timeout 200 python -m monodetr_amd.tools.wgradbench --iters 30 2>/dev/null | python -c "
import sys
for ln in sys.stdin:
    if 'kernel_ms' in ln:
        name, rest = ln.split(' ', 1); d = eval(rest); print('   %-22s lib %s kernel %s' % (name, d['library_ms'], d['kernel_ms']))"
done
ob() { local name=$1; shift
    env "$@" timeout 120 python -m monodetr_amd.tools.opbench --dtype bf16 --dist init --iters 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d['encoder']; print('%-16s enc bwd %.4f %s' % ('$name', e['bwd_ms'], e['bwd_kernels_ms']))"; }
ob base MDETR_NOOP=1
ob tile20x32 MDETR_MSDA_TILE_H=20
ob tile24x28 MDETR_MSDA_TILE_W=28
ob tile16x48 MDETR_MSDA_TILE_H=16 MDETR_MSDA_TILE_W=48
ob chunks10 MDETR_MSDA_CHUNKS=10
ob chunks14 MDETR_MSDA_CHUNKS=14
ob groups4lps8 MDETR_MSDA_LPS=8
ob base2 MDETR_NOOP=1
