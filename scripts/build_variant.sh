#!/bin/bash
# developer tool: a second build of the library with extra -D flags on ONE translation unit (timing variants inside one GPU call)
#   scripts/build_variant.sh <name> <file.hip> <flags...>   ->  monodetr_amd/variants/lib_<name>.so
set -e
R=$(cd $(dirname $0)/..; pwd); name=$1; src=$2; shift 2
mkdir -p $R/monodetr_amd/variants $R/monodetr_amd/build_obj/var
obj=$R/monodetr_amd/build_obj/var/${name}.o
cd $R/monodetr_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -Wno-pass-failed -I $R/include -I . "$@" -c $src -o $obj
# exactly the objects of the library's sources (monodetr_amd/build.py: sources()), minus the one being replaced
others=$(cd $R && python -c "
from monodetr_amd import build; import os
build.build()
print(' '.join(os.path.join(build.HERE, 'build_obj', os.path.basename(s)[:-4] + '.o') for s in build.sources() if os.path.basename(s) != '$(basename $src)'))")
hipcc --offload-arch=gfx950 -fPIC -shared -o $R/monodetr_amd/variants/lib_${name}.so $obj $others
echo built $R/monodetr_amd/variants/lib_${name}.so
