#!/bin/bash
# round 4, second GPU check: several timesteps per launch, the example-sized configurations, parked
# tables, the enqueue budget of a sharded step; then the whole suite and the bench lines.
# usage (GPU box, via gpurun): tools/r04_check2.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04b}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
( time timeout 900 python -m pytest tests -q -m gpu -x -k "batch or steps_per_launch or example_sized or availability or tail or marginal" ) > $OUT/pytest_new.log 2>&1; tail -25 $OUT/pytest_new.log
( time timeout 900 python -m pytest tests -q -m gpu --durations=8 ) > $OUT/pytest_gpu.log 2>&1; tail -20 $OUT/pytest_gpu.log
B="--no-cpu-baseline --no-materialised --no-screened"
for cfg in C1 E1 E2; do
  for k in 1 8; do
    timeout 300 python bench.py --config $cfg --steps 64 --warmup 8 --steps-per-launch $k $B > $OUT/bench_${cfg}_k$k.json 2>> $OUT/bench.err
    python - $OUT/bench_${cfg}_k$k.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], "ms/step", round(d["ms_per_step"], 4), "value %.3e" % d["value"], d["kernel"]["name"],
          "kernel ms/step", round(d["kernel"]["avg_ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 3), d["roofline"]["bound"],
          "copies", round(d.get("step_with_copies", {}).get("ms_per_step", -1), 4))
except Exception as e:
    print(sys.argv[1], "failed:", e)
PY
  done
done
tail -5 $OUT/bench.err
A=$OUT/ab.txt; : > $A
timeout 300 python tools/ab.py --config C3L --mode marginal --steps 8 --engines '[{}, {"shift_tail": 0}, {"shift": 0}]' - >> $A 2>&1
timeout 300 python tools/ab.py --config C2 --mode detect --steps 10 --engines '[{}]' - >> $A 2>&1
cat $A
timeout 300 python tools/enqueue_budget.py --world 8 --rank 3 > $OUT/enqueue_C3_rank3of8.json 2> $OUT/enqueue.err; cat $OUT/enqueue_C3_rank3of8.json; tail -2 $OUT/enqueue.err
timeout 300 python tools/enqueue_budget.py --world 8 --rank 3 --partition planes > $OUT/enqueue_C3_rank3of8_planes.json 2>> $OUT/enqueue.err; cat $OUT/enqueue_C3_rank3of8_planes.json
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench_C3.json 2>> $OUT/bench.err; python - $OUT/bench_C3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("C3 ms/step", d["ms_per_step"], "value %.4e" % d["value"], "kernel", d["kernel"])
for k in ("table_switch", "roofline_materialised", "locate_marginal", "step_with_copies"):
    print(k, json.dumps(d.get(k))[:900])
PY
tail -3 $OUT/bench.err
