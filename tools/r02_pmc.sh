#!/bin/bash
# PMC passes over three float64 kernel variants on one C3 detect step (run on the GPU box via gpurun)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02d
mkdir -p $OUT
cd $ROOT
python -m pytest tests -q -m gpu -k "paired" > $OUT/pytest_pair.log 2>&1; tail -3 $OUT/pytest_pair.log
{
python tools/ab.py --config C3 --mode detect --steps 4 --engines '[{"pair":0,"exact":1,"groups":128},{"pair":0,"exact":1,"groups":512},{"pair":0,"exact":1,"groups":1024},{"pair":0,"exact":1,"waves":16,"lds_bytes":163840},{"pair":0,"exact":0,"samples_per_lane":2,"lds_bytes":54272},{"pair":1,"groups":128},{"pair":1,"groups":512}]' -
} > $OUT/knobs.txt 2>&1; cat $OUT/knobs.txt
for variant in old exact pair; do
  case $variant in
    old) SWEEP='[{"pair":0,"exact":0}]';;
    exact) SWEEP='[{"pair":0,"exact":1}]';;
    pair) SWEEP='[{"pair":1}]';;
  esac
  bash tools/prof_counters.sh C3 "$SWEEP" r02d/pmc_$variant > $OUT/pmc_$variant.txt 2>&1
  grep -E "stack_|exact|pair" $OUT/pmc_$variant.txt | head -20
done
