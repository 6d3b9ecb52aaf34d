#!/bin/bash
# row blocks on 32-byte records (the other loops keep 64-byte ones): tests, fuzz, timings
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04h
mkdir -p $OUT
cd $ROOT
( timeout 900 python -m pytest tests -q -m gpu -x -k "row_blocks or shift or tail or batch" ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
bash tools/r04_fuzz.sh 8 300 300
for r in 66 100 128 200; do timeout 600 python tools/tune.py --config C3 --rows $r --ns 1536 --reps 2 --sweep '[{}, {"shift_rows_direct": 2}, {"shift_rows_direct": 0, "shift": 1}]' 2>&1 | grep cfg; done | tee $OUT/rows_forms_packed.txt
timeout 600 python tools/tune.py --config C3 --rows 128 --ns 401 --volume --reps 2 --sweep '[{}]' 2>&1 | grep cfg | tee -a $OUT/rows_forms_packed.txt
