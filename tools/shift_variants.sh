#!/bin/bash
# library variants that differ in the generated shift-reuse loop (constants of gen_shift_asm.py, set through
# tools/dev/shift_overlay.py) or in -D defines of qm_launch_shift.hip.
# usage: tools/shift_variants.sh name[:CONST=VAL;CONST=VAL][@-DX=1 ...] ...
# result: build_variants/libqmhip_<name>.so (the other objects are the in-tree build's)
cd "$(dirname "$0")/.."
C=quakemigrate_amd/csrc
mkdir -p build_variants
for spec in "$@"; do
  defs=""; [[ "$spec" == *@* ]] && defs="${spec#*@}" && spec="${spec%%@*}"
  name=${spec%%:*}; envs=""; [ "$spec" != "$name" ] && envs=$(echo "${spec#*:}" | tr ';' ' ')
  inc=$PWD/build_variants/shift_asm_$name.inc
  python tools/dev/shift_overlay.py $envs > $inc || exit 1
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DQM_SHIFT_ASM_INC="\"$inc\"" $defs -c $C/qm_launch_shift.hip \
      -o build_variants/qm_launch_shift_$name.o 2>&1 | grep -E "error|Spill" 
  objs=$(ls $C/build/*.o | grep -v qm_launch_shift)
  if [ "${QM_VARIANT_ENGINE:-0}" = "1" ]; then
    # knobs that change the record stream's format or the loop's constants also change the stream's
    # builder (qm_tables.hip) and the launch geometry (qm_engine.hip): both include the generated file
    for unit in qm_tables qm_engine; do
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DQM_SHIFT_ASM_INC="\"$inc\"" -c $C/$unit.hip \
          -o build_variants/${unit}_$name.o 2>&1 | grep -E "error"
      objs=$(echo "$objs" | grep -v "/$unit.hip.o")" build_variants/${unit}_$name.o"
    done
  fi
  hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/libqmhip_$name.so $objs build_variants/qm_launch_shift_$name.o || exit 1
done
ls -la build_variants/libqmhip_*.so | awk '{print $5, $9}'
