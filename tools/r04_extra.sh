#!/bin/bash
# extras of the round-4 evidence: the undecimated Askja-sized grid, and rocprofv3 kernel stats of a bench
# command whose only launches of the headline kernel are the timed steps
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04x
mkdir -p $OUT
cd $ROOT
for k in 1 8; do python bench.py --config E2F --steps 32 --warmup 4 --steps-per-launch $k --no-cpu-baseline --no-materialised --no-screened > $OUT/bench_E2F_k$k.json 2>> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_E2F_k$k.json')); print('E2F K=$k ms/step', d['ms_per_step'], d['kernel']['name'], d['kernel']['avg_ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['bound'])"; done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_C3_with_traffic.json 2>> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_C3_with_traffic.json')); print('C3', d['ms_per_step'], 'traffic', d['roofline']['traffic'], json.dumps(d['roofline']['traffic_detail'])[:400]); print(json.dumps(d['roofline_materialised'])[:600])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- \
    python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-screened --no-table-switch --no-copies --no-materialised > $OUT/bench_C3_headline_only_under_rocprof.json 2> $OUT/prof.err
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/bench_C3_headline_only_kernel_stats.csv \;
head -5 $OUT/bench_C3_headline_only_kernel_stats.csv
python -c "
import json; d=json.load(open('$OUT/bench_C3_headline_only_under_rocprof.json')); print('bench line under rocprof: kernel avg_ms', d['kernel']['avg_ms'], 'launches', d['kernel']['launches'], 'ms/step', d['ms_per_step'])"
find $OUT/prof -name "*.csv" -size +1M -delete
