#!/bin/bash
# PMC passes over one engine configuration (development aid; run on the GPU box via gpurun).
# usage: tools/prof_counters.sh <config> '<sweep json>' <tag>
set -u
CFG=${1:-C2}; SWEEP=${2:-'[{"samples_per_lane":4,"waves":8}]'}; TAG=${3:-pmc}; EXTRA=${4:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- \
      python $ROOT/tools/tune.py --config $CFG --reps 1 --sweep "$SWEEP" $EXTRA > $OUT/$name.log 2>&1
  local f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r.get("Kernel_Name", "?")
    if "stack_" not in k and "screen_" not in k:
        continue
    agg[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k, dict(v))
PY
  else
    echo "no counter csv for $name"; tail -5 $OUT/$name.log
  fi
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
# keep only the small csv summaries
find $OUT -name "*.csv" -size +2M -delete
