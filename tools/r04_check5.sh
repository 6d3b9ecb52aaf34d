#!/bin/bash
# the 4-wave row-block form beside the 8-wave ones (tables of more than 64 rows)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04e}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
( time timeout 900 python -m pytest tests -q -m gpu -x -k "row_blocks or batch or example_sized" ) > $OUT/pytest_new.log 2>&1; tail -8 $OUT/pytest_new.log
for r in 66 100 128 200; do timeout 600 python tools/tune.py --config C3 --rows $r --ns 1536 --reps 2 --sweep '[{"shift_rows_direct": 1}, {"shift_rows_direct": 2}, {"shift_rows_direct": 2, "shift_lazy": 1}, {"shift": 0}]' 2>&1 | grep cfg; done | tee $OUT/rows_forms.txt
timeout 600 python tools/tune.py --config C3 --rows 128 --ns 401 --volume --reps 2 --sweep '[{"shift_rows_direct": 1}, {"shift_rows_direct": 2}]' 2>&1 | grep cfg | tee -a $OUT/rows_forms.txt
timeout 300 python bench.py --config E2 --steps 64 --warmup 8 --no-cpu-baseline --no-materialised --no-screened --no-copies > $OUT/bench_E2.json 2>> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_E2.json')); print('E2 ms/step', d['ms_per_step'], d['kernel']['avg_ms'], d['roofline']['frac'])"
