#!/bin/bash
# 32-byte stream records (register indices as bytes): the suite's shift tests on the variant library,
# then A/B beside the in-tree build; the whole 6000-sample step materialised
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04g
mkdir -p $OUT
cd $ROOT
( QM_HIP_LIB=$ROOT/build_variants/libqmhip_packed.so timeout 900 python -m pytest tests -q -m gpu -x -k "shift or tail or marginal or batch or golden or sharded" ) > $OUT/pytest_packed.log 2>&1; tail -4 $OUT/pytest_packed.log
( QM_HIP_LIB=$ROOT/build_variants/libqmhip_packed.so timeout 600 python tools/fuzz_shift.py 300 77 ) > $OUT/fuzz_packed.log 2>&1; tail -2 $OUT/fuzz_packed.log
A=$OUT/ab.txt; : > $A
for rep in 1 2; do
timeout 300 python tools/ab.py --config C3 --mode detect --steps 8 --engines '[{}]' - build_variants/libqmhip_packed.so >> $A 2>&1
done
timeout 300 python tools/ab.py --config C1 --mode detect --steps 40 --engines '[{}]' - build_variants/libqmhip_packed.so >> $A 2>&1
timeout 300 python tools/ab.py --config C3L --mode volume --steps 8 --engines '[{}]' - build_variants/libqmhip_packed.so >> $A 2>&1
timeout 300 python tools/ab.py --config C3L --mode marginal --steps 8 --engines '[{}]' - build_variants/libqmhip_packed.so >> $A 2>&1
timeout 300 python tools/ab.py --config C3 --mode detect --steps 4 --case '{"rows": 60}' --engines '[{}]' - build_variants/libqmhip_packed.so >> $A 2>&1
timeout 300 python tools/ab.py --config C3 --mode detect --steps 3 --case '{"rows": 128, "n_samples": 1536}' --engines '[{}]' - build_variants/libqmhip_packed.so >> $A 2>&1
cat $A
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-screened --no-copies --no-table-switch --materialise-full > $OUT/bench_C3_full_volume.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_C3_full_volume.json')); print(json.dumps(d.get('roofline_materialised_full')))"; tail -2 $OUT/bench.err
