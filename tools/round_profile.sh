#!/bin/bash
# Round evidence: GPU tests, smoke, bench line, rocprofv3 kernel stats of the same bench command.
# usage (on the GPU box via gpurun): tools/round_profile.sh <round-tag>
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- \
    python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
head -8 $OUT/bench_kernel_stats.csv
find $OUT/prof -name "*.csv" -size +1M -delete
