#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02i
mkdir -p $OUT
cd $ROOT
( time timeout 1500 python -m pytest tests -q -m gpu -s -k "screen or adversarial or exp2f or c4_slab or empty_scan" ) > $OUT/pytest.log 2>&1; grep -E "worst|passed|failed|FAILED|Error|v_exp" $OUT/pytest.log | head -40
{
python tools/ab.py --config C3 --mode detect --steps 6 --engines '[{"screen":0},{"screen":1},{"screen":1,"screen_pairs":2,"screen_big":0},{"screen":1,"screen_pairs":2,"screen_big":1}]' -
python tools/ab.py --config C2 --mode detect --steps 10 --engines '[{"screen":0},{"screen":1}]' -
python tools/ab.py --config C1 --mode detect --steps 20 --engines '[{"screen":0},{"screen":1}]' -
python tools/ab.py --config C4 --mode detect --steps 3 --case '{"x_range":[150,200]}' --engines '[{"screen":0},{"screen":1}]' -
} > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
