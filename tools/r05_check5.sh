#!/bin/bash
# round 5, fifth GPU call: tie_rule tests on the fixed exp, its cost, the C1 bench line with copies again
cd "$(dirname "$0")/.."
O=gpurun_out/r05_check5; mkdir -p $O
( python -m pytest tests -m gpu -x -q -k "tie_rule or near_ties or twins or device_exp" 2>&1 | tail -8 ) > $O/new_tests.txt 2>&1
{
python tools/ab.py --config C3 --steps 6 --engines '[{}, {"tie_rule": 1}]' -
python tools/ab.py --config C1 --steps 20 --engines '[{}, {"tie_rule": 1}]' -
python tools/ab.py --config C2 --steps 10 --engines '[{}, {"tie_rule": 1}]' -
} 2>&1 | tee $O/ab.txt
for i in 1 2; do
python bench.py --config C1 --steps 400 --warmup 8 --steps-per-launch 8 --no-cpu-baseline --no-screened --no-materialised --no-table-switch 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['step_with_copies']
print('bench C1 k8: resident', round(d['ms_per_step'],4), 'with copies', round(c['ms_per_step'],4), 'ratio', round(c['ms_per_step']/d['ms_per_step'],3))"
done 2>&1 | tee $O/bench.txt
python tools/diag_stream.py C1 8 3 2>&1 | grep -v amdgpu | tee -a $O/bench.txt
