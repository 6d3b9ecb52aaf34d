#!/bin/bash
# after the host-copy change (results through the pinned bounce buffer): the lines that move results to the host
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04f2
mkdir -p $OUT
cd $ROOT
python bench.py --config C5 --steps 30 --warmup 3 > $OUT/bench_C5_stream.json 2> $OUT/bench.err
python bench.py --config C5 --steps 720 --warmup 3 --no-cpu-baseline > $OUT/bench_C5_24h.json 2>> $OUT/bench.err
for cfg in C1 E2; do python bench.py --config $cfg --steps 64 --warmup 8 --steps-per-launch 1 --no-cpu-baseline --no-materialised --no-screened > $OUT/bench_${cfg}_k1.json 2>> $OUT/bench.err; done
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "ms/step", round(d["ms_per_step"], 4), "value %.4e" % d["value"], d["kernel"]["name"], "frac", round(d["roofline"]["frac"], 3),
      "| copies", (d.get("step_with_copies") or {}).get("ms_per_step"))
PY
done
python tools/widen_bench.py > $OUT/widen_rows.jsonl 2> $OUT/widen.err; cat $OUT/widen_rows.jsonl
python tools/enqueue_budget.py --world 8 --rank 3 > $OUT/enqueue_C3_rank3of8.json 2> $OUT/enqueue.err; cat $OUT/enqueue_C3_rank3of8.json
for f in $OUT/bench.err $OUT/widen.err $OUT/enqueue.err; do tail -n 3 $f | grep -v amdgpu.ids; done; true
