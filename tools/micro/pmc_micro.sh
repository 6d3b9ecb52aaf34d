#!/bin/bash
# PMC passes (SQ / GRBM, separate runs) over a micro-benchmark binary; run on the GPU box.
# usage: tools/micro/pmc_micro.sh <tag> <kernel name substring> <binary> [args...]
set -u
TAG=$1; KSUB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
BIN=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- $BIN $ARGS > $OUT/$name.log 2>&1
  local f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$KSUB" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r.get("Kernel_Name", "?")
    if sys.argv[2] not in k: continue
    agg[k[:50]][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k[:50], r["Counter_Name"])] += 1
for k, v in agg.items():
    print(k, {c: x / n[(k, c)] for c, x in v.items()}, "dispatches", max(n.values()))
PY
}
ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
find $OUT -name "*.csv" -size +2M -delete
grep -h "kernel " $OUT/grbm.log
