#!/bin/bash
# Round 2, GPU call 6: parameter sweep of the one-pass MSDA backward at 16 waves per workgroup; where the step's ATen
# operators come from (source lines) on the GPU; KAT self checks on the device.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r02f; mkdir -p $O
cd $R
export TMPDIR=/tmp
ob() { echo "== $1 ${2:-init} ${3:-bf16}"; env $1 timeout 120 python -m monodetr_amd.tools.opbench --dist ${2:-init} --dtype ${3:-bf16} --iters 30 2>&1 | tail -1 | tee $O/opbench_$(echo "$1$2$3" | tr -c 'A-Za-z0-9\n' '_').json | python -c "
import sys, json
d = json.loads(sys.stdin.read()); e, c = d['encoder'], d['decoder']
print('   encoder bwd %.4f ms %s | decoder bwd %.4f ms %s' % (e['bwd_ms'], e['bwd_kernels_ms'], c['bwd_ms'], c['bwd_kernels_ms']))"; }
T="MDETR_MSDA_THREADS=1024"
ob "$T"
ob "$T MDETR_MSDA_CHUNKS=16"
ob "$T MDETR_MSDA_CHUNKS=12"
ob "$T MDETR_MSDA_TILE_H=20 MDETR_MSDA_TILE_W=32"
ob "$T MDETR_MSDA_TILE_H=20 MDETR_MSDA_TILE_W=32 MDETR_MSDA_CHUNKS=16"
ob "$T MDETR_MSDA_TILE_H=16 MDETR_MSDA_TILE_W=40 MDETR_MSDA_CHUNKS=16"
ob "$T MDETR_MSDA_TILE_H=12 MDETR_MSDA_TILE_W=32 MDETR_MSDA_CHUNKS=16"
ob "$T MDETR_MSDA_REACH=4 MDETR_MSDA_CHUNKS=16"
ob "$T MDETR_MSDA_CHUNKS=16" trained
ob "MDETR_MSDA_GROUPS=8 MDETR_MSDA_TILE_H=24 MDETR_MSDA_TILE_W=40 MDETR_MSDA_CHUNKS=16"
ob "MDETR_MSDA_GROUPS=4 MDETR_MSDA_CHUNKS=16" init fp32
ob "MDETR_MSDA_GROUPS=4 MDETR_MSDA_TILE_H=24 MDETR_MSDA_TILE_W=40 MDETR_MSDA_CHUNKS=16" init fp32
ob "$T MDETR_MSDA_CHUNKS=16 --hires-placeholder" init bf16 2>/dev/null | head -0
echo "== hires (config 5 shapes)"; MDETR_MSDA_THREADS=1024 MDETR_MSDA_CHUNKS=16 timeout 120 python -m monodetr_amd.tools.opbench --dist init --dtype bf16 --iters 20 --hires 2>&1 | tail -1 | cut -c1-700
timeout 300 python tests/opcount.py --device cuda --precision bf16 --top 45 --op "_to_copy,add.Tensor,copy_,clone,zeros,fill_,sum.dim_IntList,mul.Tensor" > $O/opcount_cuda_bf16.txt 2>&1; head -150 $O/opcount_cuda_bf16.txt
timeout 200 python -m pytest tests/test_fused_gpu.py -q -p no:cacheprovider -k "kitti_preprocess or device_loader or rotated_overlap or official_evaluation" 2>&1 | tail -3
