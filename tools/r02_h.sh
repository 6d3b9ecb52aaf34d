#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02h
mkdir -p $OUT
cd $ROOT
( time timeout 1500 python -m pytest tests -q -m gpu -k "paired or c3_locate or reference_signature or fused_detect or compute or marginal" ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
{
python tools/ab.py --config C3L --mode volume --engines '[{"pair":0},{"pair":1}]' -
python tools/ab.py --config C3L --mode volume --case '{"n_samples":200}' --engines '[{"pair":0},{"pair":1},{"pair":2}]' -
python tools/ab.py --config C3L --mode volume --case '{"n_samples":1001}' --engines '[{"pair":0},{"pair":1}]' -
} > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
bash tools/prof_counters.sh C3 '[{"pair":1}]' r02h/pmc_pairvol "--ns 401 --volume" > $OUT/pmc_pairvol.txt 2>&1; grep -E "stack_" $OUT/pmc_pairvol.txt | head
