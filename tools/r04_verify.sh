#!/bin/bash
# the round's closing check on the final code: the whole GPU suite, smoke(), the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04v
mkdir -p $OUT
cd $ROOT
( time python -m pytest tests -q -m gpu --durations=5 ) > $OUT/pytest_gpu.log 2>&1; tail -10 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_C3.json 2> $OUT/bench.err; python - $OUT/bench_C3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("C3 ms/step", d["ms_per_step"], "value %.4e" % d["value"], "frac", round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"],
      "| copies", d["step_with_copies"]["ms_per_step"], "| marginal", d["locate_marginal"]["avg_ms"], "| materialised", d["roofline_materialised"]["frac"],
      "| switch", d["table_switch"]["rebuild_ms"], d["table_switch"]["table_switch_ms"], "| cpu", d["cpu_baseline"]["value"])
PY
tail -2 $OUT/bench.err
