#!/bin/bash
# round-4 evidence: full GPU suite, smoke, the bench lines of every configuration (C3 headline, the
# round-2 / round-3 forms beside it, C1 / E1 / E2 with 1 and 8 timesteps per launch, C2, a C4 slab, the
# C5 stream incl. the literal 720-step day), N > 1 plumbing on one GPU, the enqueue budget of a sharded
# step, rocprofv3 kernel stats of the bench command, PMC passes over the detect, volume and marginal-map
# launches, the rows SURVEY 8(f) marks next.
# usage (GPU box, via gpurun): tools/r04_final.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
( time python -m pytest tests -q -m gpu --durations=6 ) > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py --steps 20 --warmup 3 > $OUT/bench_C3.json 2> $OUT/bench.err; tail -c 1200 $OUT/bench_C3.json; tail -3 $OUT/bench.err
Q="--no-cpu-baseline --no-screened --no-copies --no-table-switch"
python bench.py --steps 20 --warmup 3 --engine '{"shift_tail": 0}' $Q > $OUT/bench_C3_round3_tiles.json 2>> $OUT/bench.err
python bench.py --steps 20 --warmup 3 --engine '{"shift": 0}' $Q > $OUT/bench_C3_round2_kernels.json 2>> $OUT/bench.err
python bench.py --config C2 --steps 30 --warmup 3 --no-cpu-baseline --no-materialised > $OUT/bench_C2.json 2>> $OUT/bench.err
for cfg in C1 E1 E2; do for k in 1 8; do
  python bench.py --config $cfg --steps 64 --warmup 8 --steps-per-launch $k --no-cpu-baseline --no-materialised --no-screened > $OUT/bench_${cfg}_k$k.json 2>> $OUT/bench.err
done; done
python bench.py --config C4 --emulate-world 8 --emulate-rank 3 --steps 5 --warmup 1 --no-cpu-baseline --no-materialised > $OUT/bench_C4_slab3of8.json 2>> $OUT/bench.err
python bench.py --config C5 --steps 30 --warmup 3 > $OUT/bench_C5_stream.json 2>> $OUT/bench.err
python bench.py --config C5 --steps 720 --warmup 3 > $OUT/bench_C5_24h.json 2>> $OUT/bench.err; tail -c 300 $OUT/bench_C5_24h.json
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], "ms/step", round(d["ms_per_step"], 4), "value %.4e" % d["value"], d["kernel"]["name"],
          "frac", round(d["roofline"]["frac"], 3), d["roofline"]["bound"])
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
# N > 1 plumbing on one GPU (gloo rendezvous; RCCL refuses two ranks on one device): self-launch
QM_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 1 > $OUT/bench_C3_2ranks_one_gpu.json 2> $OUT/bench_2ranks.err; tail -c 300 $OUT/bench_C3_2ranks_one_gpu.json; tail -2 $OUT/bench_2ranks.err
QM_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --config C5 --steps 6 --warmup 1 > $OUT/bench_C5_2ranks_one_gpu.json 2>> $OUT/bench_2ranks.err
QM_BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-screened --no-copies --no-materialised --no-table-switch > $OUT/bench_C3_one_rank_rccl.json 2>> $OUT/bench_2ranks.err; tail -c 300 $OUT/bench_C3_one_rank_rccl.json
python tools/enqueue_budget.py --world 8 --rank 3 > $OUT/enqueue_C3_rank3of8.json 2> $OUT/enqueue.err; cat $OUT/enqueue_C3_rank3of8.json
python tools/enqueue_budget.py --world 8 --rank 3 --partition planes > $OUT/enqueue_C3_rank3of8_planes.json 2>> $OUT/enqueue.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- \
    python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_C3_under_rocprof.json 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/bench_C3_kernel_stats.csv \;
head -8 $OUT/bench_C3_kernel_stats.csv
find $OUT/prof -name "*.csv" -size +1M -delete
cd $ROOT
bash tools/prof_counters.sh C3 '[{}]' $TAG/pmc_shift > $OUT/pmc_shift.txt 2>&1; grep -E "stack_" $OUT/pmc_shift.txt | head
bash tools/prof_counters.sh C3 '[{}]' $TAG/pmc_locate "--ns 401 --volume" > $OUT/pmc_locate.txt 2>&1; grep -E "stack_" $OUT/pmc_locate.txt | head
bash tools/prof_counters.sh C3 '[{}]' $TAG/pmc_marginal "--ns 401 --marginal" > $OUT/pmc_marginal.txt 2>&1; grep -E "stack_" $OUT/pmc_marginal.txt | head
f() { find $OUT/$1/$2 -name "*counter_collection.csv" | head -1; }
python tools/pmc_traffic.py "C3:detect=$(f pmc_shift fetch),$(f pmc_shift write)" "C3L:volume=$(f pmc_locate fetch),$(f pmc_locate write)" "C3L:marginal=$(f pmc_marginal fetch),$(f pmc_marginal write)" > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err; cat $OUT/pmc_traffic.json | head -40
python tools/widen_bench.py > $OUT/widen_rows.jsonl 2> $OUT/widen.err; cat $OUT/widen_rows.jsonl
