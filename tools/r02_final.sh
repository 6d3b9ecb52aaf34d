#!/bin/bash
# round-2 evidence: full GPU suite, smoke, bench lines, rocprof stats, PMC passes of the final kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
bash tools/round_profile.sh $TAG
cd $ROOT
bash tools/prof_counters.sh C3 '[{}]' $TAG/pmc_f64 > $OUT/pmc_f64.txt 2>&1; grep -E "stack_|exact" $OUT/pmc_f64.txt | head
bash tools/prof_counters.sh C3 '[{"screen":1}]' $TAG/pmc_screen > $OUT/pmc_screen.txt 2>&1; grep -E "screen_lds" $OUT/pmc_screen.txt | head
bash tools/prof_counters.sh C3 '[{}]' $TAG/pmc_locate "--ns 401 --volume" > $OUT/pmc_locate.txt 2>&1; grep -E "stack_" $OUT/pmc_locate.txt | head
python tools/widen_bench.py > $OUT/widen_rows.jsonl 2> $OUT/widen.err; cat $OUT/widen_rows.jsonl
