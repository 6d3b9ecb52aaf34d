#!/bin/bash
# sweep of workgroup rounds on the example-sized configurations (K = 1 and 8)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04d}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
B="--no-cpu-baseline --no-materialised --no-screened --no-copies"
for cfg in E2 E1 C1 C2; do
  for r in 2 3 4 6 12; do
    for k in 1 8; do
      timeout 300 python bench.py --config $cfg --steps 64 --warmup 8 --steps-per-launch $k $B --engine "{\"rounds\": $r}" > $OUT/b.json 2>> $OUT/bench.err
      python - $OUT/b.json $cfg $r $k <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], "rounds", sys.argv[3], "K", sys.argv[4], "ms/step", round(d["ms_per_step"], 4), "kernel ms/step", round(d["kernel"]["avg_ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 3), flush=True)
except Exception as e:
    print(sys.argv[2:], "failed:", e)
PY
    done
  done
done 2>&1 | tee $OUT/rounds_sweep.txt
tail -3 $OUT/bench.err
