#!/bin/bash
# round-2 third GPU call: paired kernel -- parity tests, then A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02c
mkdir -p $OUT
cd $ROOT
( time timeout 1500 python -m pytest tests -q -m gpu -k "paired or full_size or c4_slab or c3_locate or reference_signature or engine_configurations" --durations=5 ) > $OUT/pytest_pair.log 2>&1; tail -30 $OUT/pytest_pair.log
{
python tools/ab.py --config C3 --mode detect --engines '[{"pair":0,"exact":0},{"pair":0,"exact":1},{"pair":1}]' -
python tools/ab.py --config C3L --mode volume --engines '[{"pair":0},{"pair":1}]' -
python tools/ab.py --config C2 --mode detect --engines '[{"pair":0},{"pair":1}]' -
python tools/ab.py --config C1 --mode detect --steps 20 --engines '[{"pair":0},{"pair":1},{"pair":2}]' -
python tools/ab.py --config C4 --mode detect --steps 3 --case '{"x_range":[150,200]}' --engines '[{"pair":0},{"pair":1}]' -
} > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
