#!/bin/bash
# round 4, third GPU check: marginal butterfly, serving kernel, lazy round-2 offsets, batch groups
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04c}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
( time timeout 900 python -m pytest tests -q -m gpu -x -k "marginal or serving or example_sized or batch or tail or locate" ) > $OUT/pytest_new.log 2>&1; tail -8 $OUT/pytest_new.log
( time timeout 900 python -m pytest tests -q -m gpu --durations=5 ) > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log
A=$OUT/ab.txt; : > $A
timeout 300 python tools/ab.py --config C3L --mode marginal --steps 8 --engines '[{}, {"shift_tail": 0}]' - build_variants/libqmhip_nobfly.so >> $A 2>&1
timeout 300 python tools/ab.py --config C3L --mode volume --steps 8 --engines '[{}]' - >> $A 2>&1
cat $A
B="--no-cpu-baseline --no-materialised --no-screened"
for cfg in C1 E1 E2; do
  for k in 1 8 16; do
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
timeout 300 python tools/widen_bench.py > $OUT/widen_rows.jsonl 2> $OUT/widen.err; cat $OUT/widen_rows.jsonl; tail -2 $OUT/widen.err
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_C3.json 2>> $OUT/bench.err; python - $OUT/bench_C3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("C3 ms/step", d["ms_per_step"], "value %.4e" % d["value"], "frac", d["roofline"]["frac"], "adds_only", d["roofline"]["adds_only_frac"], "traffic", d["roofline"]["traffic"])
for k in ("table_switch", "roofline_materialised", "locate_marginal"):
    print(k, json.dumps(d.get(k))[:700])
PY
tail -3 $OUT/bench.err
