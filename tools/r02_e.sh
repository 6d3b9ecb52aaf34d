#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02e
mkdir -p $OUT
cd $ROOT
( time timeout 1500 python -m pytest tests -q -m gpu -s -k "paired or adversarial or c3_locate" ) > $OUT/pytest.log 2>&1; grep -E "worst|passed|failed|FAILED|Error" $OUT/pytest.log | head -40
{
python tools/ab.py --config C3 --mode detect --steps 4 --engines '[{"rounds":12},{"rounds":24},{"rounds":48},{"rounds":96},{"rounds":192}]' -
python tools/ab.py --config C3L --mode volume --engines '[{"pair":0},{"pair":1},{"pair":1,"rounds":48}]' -
python tools/ab.py --config C3L --mode volume --case '{"n_samples":512}' --engines '[{"pair":0},{"pair":1}]' -
python tools/ab.py --config C3L --mode volume --case '{"n_samples":256}' --engines '[{"pair":0},{"pair":1}]' -
python tools/ab.py --config C2 --mode detect --engines '[{"rounds":12},{"rounds":48}]' -
python tools/ab.py --config C4 --mode detect --steps 3 --case '{"x_range":[150,200]}' --engines '[{"rounds":12},{"rounds":48}]' -
} > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
