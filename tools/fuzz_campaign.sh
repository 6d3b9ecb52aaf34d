#!/bin/bash
# randomised differential campaign on the round's code (tools/fuzz_shift.py), seeds in parallel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-8}; TRIALS=${2:-200}; BASE=${3:-100}
OUT=$ROOT/gpurun_out/fuzz
mkdir -p $OUT
cd $ROOT
pids=()
for i in $(seq 1 $N); do
  ( timeout 1500 python tools/fuzz_shift.py $TRIALS $((BASE + i)) > $OUT/seed_$((BASE + i)).log 2>&1; echo "seed $((BASE + i)): rc=$? $(tail -1 $OUT/seed_$((BASE + i)).log | cut -c1-300)" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
grep -L "trials ok" $OUT/seed_*.log | while read f; do echo "== FAILED $f"; tail -25 $f; done
