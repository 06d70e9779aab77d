#!/bin/bash
# Round 3, call 25: conv_taps with conflict-free staging stores against the previous mapping (alternate library): op level, counters, step.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03t2; mkdir -p $O
cd $R
export TMPDIR=/tmp
show() { python -c "
import json,sys; d=json.load(open('$1')); print('$2', {k.replace('_kernel',''): v['ms'] for k, v in d.items() if k.endswith('_kernel') and ('$3' in k)})"; }
b() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants 2>$O/bench_$tag.err | tail -1 > $O/bench_$tag.json
  python -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d.get('final_loss'), d['config']['launch'])" || tail -3 $O/bench_$tag.err; }
timeout 300 python -m monodetr_amd.tools.convbench --only strided --iters 20 2>/dev/null | tail -1 > $O/strided_new.json; show $O/strided_new.json new fwd_; show $O/strided_new.json new dgrad_
timeout 300 python -m pytest tests/test_fused_gpu.py -x -q -m gpu -p no:cacheprovider -k "conv" 2>&1 | tail -2
b new X=1
cd /tmp; PYTHONPATH=$R timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_taps_new -- python -m monodetr_amd.tools.convbench --only strided --iters 2 > $O/pmc_taps_new.log 2>&1
cd $R; python -m monodetr_amd.tools.pmc_summary /tmp/pmc_taps_new --match conv --out $O/r03_pmc_conv_strided_remap.json > /dev/null 2>$O/pmc_summary.err
python -c "
import json
for r in json.load(open('$O/r03_pmc_conv_strided_remap.json')):
    if 'mdetr' in r['kernel'] or 'dgrad4' in r['kernel']: print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k in ('kernel', 'grid', 'lds_conflict_frac', 'frac_issuing')})" | cut -c1-200
cp monodetr_amd/libmonodetr_amd.so /tmp/lib_main.so; cp monodetr_amd/libmonodetr_amd_alt.so monodetr_amd/libmonodetr_amd.so
timeout 300 python -m monodetr_amd.tools.convbench --only strided --iters 20 2>/dev/null | tail -1 > $O/strided_old.json; show $O/strided_old.json old fwd_; show $O/strided_old.json old dgrad_
b old X=1
cp /tmp/lib_main.so monodetr_amd/libmonodetr_amd.so
