#!/bin/bash
# round 4, first GPU check: the tail tiles and the marginal-map flavour of the shift-reuse kernel --
# their tests first, then the whole suite, then A/B lines beside the round-3 forms.
# usage (GPU box, via gpurun): tools/r04_check1.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04a}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
( time timeout 600 python -m pytest tests -q -m gpu -x -k "tail or marginal_map_on_whole" ) > $OUT/pytest_new.log 2>&1; tail -25 $OUT/pytest_new.log
( time timeout 900 python -m pytest tests -q -m gpu --durations=8 ) > $OUT/pytest_gpu.log 2>&1; tail -30 $OUT/pytest_gpu.log
A=$OUT/ab.txt; : > $A
timeout 300 python tools/ab.py --config C3 --mode detect --steps 8 --engines '[{}, {"shift_tail": 0}]' - >> $A 2>&1
timeout 300 python tools/ab.py --config C3L --mode volume --steps 8 --engines '[{}, {"shift_tail": 0}, {"shift": 0}]' - >> $A 2>&1
timeout 300 python tools/ab.py --config C3L --mode marginal --steps 8 --engines '[{}, {"shift_tail": 0}, {"shift": 0}]' - >> $A 2>&1
timeout 300 python tools/ab.py --config C1 --mode detect --steps 40 --engines '[{}, {"shift_tail": 0}]' - >> $A 2>&1
cat $A
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench_C3.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench_C3.json; tail -3 $OUT/bench.err
