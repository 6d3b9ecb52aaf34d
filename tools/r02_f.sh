#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02f
mkdir -p $OUT
cd $ROOT
( time timeout 1500 python -m pytest tests -q -m gpu -k "paired or c3_locate or full_size or engine_configurations or any_row_count" ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
{
python tools/ab.py --config C3 --mode detect --steps 5 --engines '[{"exact":0},{"exact":1},{"pair":2}]' -
python tools/ab.py --config C3L --mode volume --engines '[{"pair":0},{"pair":1}]' -
python tools/ab.py --config C2 --mode detect --engines '[{"exact":0},{"exact":1}]' -
python tools/ab.py --config C1 --mode detect --steps 20 --engines '[{"exact":0},{"exact":1}]' -
python tools/ab.py --config C4 --mode detect --steps 3 --case '{"x_range":[150,200]}' --engines '[{"exact":0},{"exact":1}]' -
} > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
bash tools/prof_counters.sh C3 '[{"pair":1}]' r02f/pmc_pairvol "--ns 401 --volume" > $OUT/pmc_pairvol.txt 2>&1; grep -E "stack_" $OUT/pmc_pairvol.txt | head
