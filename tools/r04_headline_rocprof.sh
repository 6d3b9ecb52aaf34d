set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- \
    python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-screened --no-table-switch --no-copies --no-materialised > $OUT/bench_C3_headline_only_under_rocprof.json 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/bench_C3_headline_only_kernel_stats.csv \;
head -4 $OUT/bench_C3_headline_only_kernel_stats.csv
python -c "
import json; d=json.load(open('$OUT/bench_C3_headline_only_under_rocprof.json')); print('bench line under rocprof: kernel avg_ms', d['kernel']['avg_ms'], 'launches', d['kernel']['launches'], 'ms/step', d['ms_per_step'])"
find $OUT/prof -name "*.csv" -size +1M -delete
