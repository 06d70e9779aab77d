#!/bin/bash
# build and run scripts/exp/conv_timeline.hip on the GPU box:  bash scripts/exp/conv_timeline.sh <tag> "<tune>;<tune>" C H W
T=${1:-r06tl}; V=${2:--}; shift 2
O=gpurun_out/$T; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -Wno-pass-failed -Wno-unused-value -I include -I monodetr_amd/csrc \
    scripts/exp/conv_timeline.hip -L monodetr_amd -lmonodetr_amd -Wl,-rpath,$PWD/monodetr_amd -o /tmp/conv_timeline_$$ 2> $O/build.err || { tail -5 $O/build.err; exit 1; }
IFS=';' read -ra VS <<< "$V"
for v in "${VS[@]}"; do
    name=$(echo "$v" | tr ',=' '__'); [ "$v" = "-" ] && v="" && name=default
    MDETR_TUNE="$v" /tmp/conv_timeline_$$ "$@" > $O/timeline_$name.txt 2>&1
    echo "== $name"; head -7 $O/timeline_$name.txt; grep -A8 "^SPAN" $O/timeline_$name.txt | cut -c1-600
done
rm -f /tmp/conv_timeline_$$
