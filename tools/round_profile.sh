#!/bin/bash
# Round evidence: GPU tests, smoke, bench lines, rocprofv3 kernel stats of the same bench command,
# a two-rank run of the N>1 path on one GPU.
# usage (on the GPU box via gpurun): tools/round_profile.sh <round-tag>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
( time python -m pytest tests -q -m gpu --durations=6 ) > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py --steps 20 --warmup 3 > $OUT/bench_C3.json 2> $OUT/bench.err; tail -c 2500 $OUT/bench_C3.json; tail -3 $OUT/bench.err
python bench.py --config C2 --steps 30 --warmup 3 --no-cpu-baseline --no-materialised > $OUT/bench_C2.json 2>> $OUT/bench.err
python bench.py --config C1 --steps 50 --warmup 5 --no-cpu-baseline --no-materialised > $OUT/bench_C1.json 2>> $OUT/bench.err
python bench.py --config C4 --emulate-world 8 --emulate-rank 3 --steps 5 --warmup 1 --no-cpu-baseline --no-materialised > $OUT/bench_C4_slab3of8.json 2>> $OUT/bench.err
python bench.py --config C5 --steps 30 --warmup 3 > $OUT/bench_C5_stream.json 2>> $OUT/bench.err
# N > 1 plumbing on one GPU (gloo rendezvous; RCCL refuses two ranks on one device): self-launch
QM_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 1 > $OUT/bench_C3_2ranks_one_gpu.json 2> $OUT/bench_2ranks.err; tail -c 600 $OUT/bench_C3_2ranks_one_gpu.json; tail -3 $OUT/bench_2ranks.err
QM_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --config C4 --steps 2 --warmup 1 > $OUT/bench_C4_2ranks_one_gpu.json 2>> $OUT/bench_2ranks.err; tail -c 400 $OUT/bench_C4_2ranks_one_gpu.json
QM_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --config C5 --steps 6 --warmup 1 > $OUT/bench_C5_2ranks_one_gpu.json 2>> $OUT/bench_2ranks.err; tail -c 400 $OUT/bench_C5_2ranks_one_gpu.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- \
    python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_C3_under_rocprof.json 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/bench_C3_kernel_stats.csv \;
head -8 $OUT/bench_C3_kernel_stats.csv
find $OUT/prof -name "*.csv" -size +1M -delete
