#!/bin/bash
# round-3 evidence: full GPU suite, smoke, bench lines (C3 headline, C1, C2, C4 slab, C5 stream incl. the
# literal 720-step day), N > 1 plumbing on one GPU, rocprofv3 kernel stats of the bench command, PMC
# passes over the shift-reuse kernel (detect and locate window) and the round-2 kernel beside it.
# usage (GPU box, via gpurun): tools/r03_final.sh [tag]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
( time python -m pytest tests -q -m gpu --durations=6 ) > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py --steps 20 --warmup 3 > $OUT/bench_C3.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench_C3.json; tail -3 $OUT/bench.err
python bench.py --steps 20 --warmup 3 --engine '{"shift": 0}' --no-cpu-baseline --no-screened --no-copies > $OUT/bench_C3_round2_kernels.json 2>> $OUT/bench.err
python bench.py --config C2 --steps 30 --warmup 3 --no-cpu-baseline --no-materialised > $OUT/bench_C2.json 2>> $OUT/bench.err
python bench.py --config C1 --steps 50 --warmup 5 --no-cpu-baseline --no-materialised > $OUT/bench_C1.json 2>> $OUT/bench.err
python bench.py --config C4 --emulate-world 8 --emulate-rank 3 --steps 5 --warmup 1 --no-cpu-baseline --no-materialised > $OUT/bench_C4_slab3of8.json 2>> $OUT/bench.err
python bench.py --config C5 --steps 30 --warmup 3 > $OUT/bench_C5_stream.json 2>> $OUT/bench.err
python bench.py --config C5 --steps 720 --warmup 3 > $OUT/bench_C5_24h.json 2>> $OUT/bench.err; tail -c 400 $OUT/bench_C5_24h.json
# N > 1 plumbing on one GPU (gloo rendezvous; RCCL refuses two ranks on one device): self-launch
QM_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 1 > $OUT/bench_C3_2ranks_one_gpu.json 2> $OUT/bench_2ranks.err; tail -c 400 $OUT/bench_C3_2ranks_one_gpu.json; tail -2 $OUT/bench_2ranks.err
QM_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --partition planes --steps 5 --warmup 1 > $OUT/bench_C3_2ranks_planes_one_gpu.json 2>> $OUT/bench_2ranks.err
QM_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --config C5 --steps 6 --warmup 1 > $OUT/bench_C5_2ranks_one_gpu.json 2>> $OUT/bench_2ranks.err
QM_BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-screened --no-copies --no-materialised > $OUT/bench_C3_one_rank_rccl.json 2>> $OUT/bench_2ranks.err; tail -c 300 $OUT/bench_C3_one_rank_rccl.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- \
    python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_C3_under_rocprof.json 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/bench_C3_kernel_stats.csv \;
head -8 $OUT/bench_C3_kernel_stats.csv
find $OUT/prof -name "*.csv" -size +1M -delete
cd $ROOT
bash tools/prof_counters.sh C3 '[{}]' $TAG/pmc_shift > $OUT/pmc_shift.txt 2>&1; grep -E "stack_" $OUT/pmc_shift.txt | head
bash tools/prof_counters.sh C3 '[{"shift": 0}]' $TAG/pmc_round2 > $OUT/pmc_round2.txt 2>&1; grep -E "stack_" $OUT/pmc_round2.txt | head
bash tools/prof_counters.sh C3 '[{}]' $TAG/pmc_locate "--ns 401 --volume" > $OUT/pmc_locate.txt 2>&1; grep -E "stack_" $OUT/pmc_locate.txt | head
# locate window on tables of 33-64 rows: the 8-wave volume variant beside the round-2 exact-row kernel
python tools/tune.py --config C3 --rows 60 --ns 401 --volume --reps 3 --sweep '[{}, {"shift": 0}]' > $OUT/locate_60rows.txt 2>&1; tail -4 $OUT/locate_60rows.txt
# tables of more than 64 rows: row blocks (LDS-direct form, register-staged form) beside the chunked kernel (C3 grid, 1536 samples)
for r in 66 128 200; do python tools/tune.py --config C3 --rows $r --ns 1536 --reps 2 --sweep '[{}, {"shift_rows_direct": 0, "shift": 1}, {"shift": 0}]' 2>&1 | grep cfg; done > $OUT/rows_66_128_200.txt; cat $OUT/rows_66_128_200.txt
python tools/tune.py --config C3 --rows 128 --ns 401 --volume --reps 2 --sweep '[{}, {"shift": 0}]' 2>&1 | grep cfg >> $OUT/rows_66_128_200.txt
python tools/widen_bench.py > $OUT/widen_rows.jsonl 2> $OUT/widen.err; cat $OUT/widen_rows.jsonl
