#!/bin/bash
# builds the micro-model against variants of the generated loop (generator constants via tools/dev/shift_overlay.py)
# CXXDEFS="-DNWAVES=6 -DWPE=3" selects the workgroup shape
# usage: tools/micro/build_shift_variants.sh name[:ENV=VAL,ENV=VAL] ...
cd "$(dirname "$0")/../.."
for spec in "$@"; do
  name=${spec%%:*}; envs=""; [ "$spec" != "$name" ] && envs=$(echo "${spec#*:}" | tr ';' ' ')
  d=build_variants/inc_$name; mkdir -p $d
  python tools/dev/shift_overlay.py $envs > $d/qm_shift_asm.inc || exit 1
  hipcc --offload-arch=gfx950 -O3 $CXXDEFS -I$d -o build_variants/shift_model_$name tools/micro/shift_model.hip 2>&1 \
    | grep -v "inline asm clobber\|Reserved registers\|1 warning generated" 
done
