#!/bin/bash
# conv3x3 (csrc/conv3x3.hip) on the GPU: parity tests, then kernel-only timings at the four ResNet stages for a list of MDETR_TUNE settings.
#   bash scripts/r06_conv.sh <tag> "<tune1>;<tune2>;..."        ("-" = no tune)
T=${1:-r06conv}; V=${2:--}
O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "conv3x3 or Conv3x3 or backbone" > $O/pytest_conv.log 2>&1; echo "pytest rc=$?" ; tail -3 $O/pytest_conv.log
IFS=';' read -ra VS <<< "$V"
for v in "${VS[@]}"; do
    name=$(echo "$v" | tr ',=' '__'); [ "$v" = "-" ] && v="" && name=default
    MDETR_TUNE="$v" python -m monodetr_amd.tools.convbench --only conv3x3 --iters 50 > $O/convbench_$name.json 2> $O/convbench_$name.err
    echo "== $name"; python - <<P
import json
j=json.load(open("$O/convbench_$name.json"))
print("  ".join("%s %.1fus %.3f" % (k.replace("conv3x3_","").replace("_kernel",""), v["ms"]*1e3, v["frac_mfma"]) for k,v in j.items() if k.startswith("conv3x3")))
P
done
