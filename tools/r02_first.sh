#!/bin/bash
# round-2 first GPU call: parity suite, micro-benchmarks, the bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $ROOT
( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 ) > $OUT/pytest_gpu.log 2>&1; tail -25 $OUT/pytest_gpu.log
( cd tools/micro && hipcc --offload-arch=gfx950 -O3 -o f64_lds f64_lds.hip && timeout 300 ./f64_lds ) > $OUT/f64_lds.txt 2>&1; cat $OUT/f64_lds.txt
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; tail -c 4000 $OUT/bench.json; tail -5 $OUT/bench.err
