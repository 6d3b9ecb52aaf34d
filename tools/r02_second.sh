#!/bin/bash
# round-2 second GPU call: parity suite (all tests), A/B of the exact-row-count kernel and exp2 degrees
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02b
mkdir -p $OUT
cd $ROOT
( time timeout 1500 python -m pytest tests -q -m gpu --durations=8 ) > $OUT/pytest_gpu.log 2>&1; tail -30 $OUT/pytest_gpu.log
{
python tools/ab.py --config C3 --mode detect --engines '[{"screen":0}]' build_variants/r01.so
python tools/ab.py --config C3 --mode detect --engines '[{"exact":0},{"exact":1}]' -
python tools/ab.py --config C3 --mode detect --engines '[{"exact":1}]' build_variants/d7.so build_variants/d6.so
python tools/ab.py --config C3L --mode volume --engines '[{"screen":0}]' build_variants/r01.so
python tools/ab.py --config C3L --mode volume --engines '[{"exact":0},{"exact":1}]' - build_variants/d7.so
python tools/ab.py --config C3L --mode marginal --engines '[{"exact":0}]' -
python tools/ab.py --config C2 --mode detect --engines '[{"exact":0},{"exact":1}]' -
python tools/ab.py --config C1 --mode detect --steps 20 --engines '[{"exact":0},{"exact":1}]' -
python tools/ab.py --config C4 --mode detect --steps 3 --case '{"x_range":[150,200]}' --engines '[{"exact":0},{"exact":1}]' -
} > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
