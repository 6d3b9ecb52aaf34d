#!/bin/bash
# round 5, fourth GPU call: tie_rule tests, its cost at C3 / C1, where the streaming pipeline's overhead goes
cd "$(dirname "$0")/.."
O=gpurun_out/r05_check4; mkdir -p $O
( python -m pytest tests -m gpu -x -q -k "tie_rule or near_ties or twins" 2>&1 | tail -8 ) > $O/new_tests.txt 2>&1
{
python tools/ab.py --config C3 --steps 6 --engines '[{}, {"tie_rule": 1}]' -
python tools/ab.py --config C1 --steps 20 --engines '[{}, {"tie_rule": 1}]' -
python tools/ab.py --config C2 --steps 10 --engines '[{}, {"tie_rule": 1}]' -
} 2>&1 | tee $O/ab.txt
{
python tools/diag_stream.py C1 8 3
python tools/diag_stream.py C1 8 4
python tools/diag_stream.py C1 1 3
python tools/diag_stream.py C1 1 6
python tools/diag_stream.py E2 1 3
python tools/diag_stream.py C1 16 3
} 2>&1 | tee $O/stream.txt
