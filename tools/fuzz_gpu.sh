#!/bin/bash
# larger randomised campaigns of the differential tests (development aid; run via gpurun)
set -u
N=${1:-12}; TRIALS=${2:-2500}; TIES=${3:-400}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
fail=0
for seed in $(seq 1 $N); do
  out=$(QM_FUZZ_TRIALS=$TRIALS QM_FUZZ_SEED=$((9000 + seed)) QM_TIES_TRIALS=$TIES QM_TIES_SEED=$((500 + seed)) \
    timeout 900 python -m pytest tests -q -m gpu -x -k "randomised or paired_kernel_every" 2>&1)
  line=$(echo "$out" | grep -E "passed|failed|error" | tail -1)
  echo "seed $seed: $line"
  echo "$line" | grep -q "failed\|error" && { fail=1; echo "$out" | grep -E "^E " | head -20; }
done
echo "fuzz done, seeds=$N trials=$TRIALS ties=$TIES fail=$fail"
