#!/bin/bash
# build and run scripts/exp/tgemm_timeline.hip on the GPU box:  bash scripts/exp/tgemm_timeline.sh <tag> T K N
T=${1:-r06tg}; shift
O=gpurun_out/$T; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -Wno-pass-failed -Wno-unused-value -I include -I monodetr_amd/csrc \
    scripts/exp/tgemm_timeline.hip -L monodetr_amd -lmonodetr_amd -Wl,-rpath,$PWD/monodetr_amd -o /tmp/tgemm_timeline_$$ 2> $O/build.err || { tail -5 $O/build.err; exit 1; }
/tmp/tgemm_timeline_$$ "$@" | tee $O/tgemm_timeline.txt
rm -f /tmp/tgemm_timeline_$$
