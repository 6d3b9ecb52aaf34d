#!/bin/bash
# timing experiments (wrong results by design): the detect loop without its per-row scalar loads / waits
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r04i
for rep in 1 2; do
timeout 300 python tools/ab.py --config C3 --mode detect --steps 8 --engines '[{}]' - build_variants/libqmhip_nosmem.so build_variants/libqmhip_nowait.so build_variants/libqmhip_packed.so build_variants/libqmhip_packednosmem.so
done 2>&1 | tee gpurun_out/r04i/ab_smem.txt
