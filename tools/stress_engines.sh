#!/bin/bash
# tools/stress_engines.py in 16 concurrent processes: modes "m:count ..." (default: the four modes, four each)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
S=${1:-90}; shift
MODES=${*:-0:4 1:4 2:4 3:4}
mkdir -p gpurun_out/stress; rm -f gpurun_out/stress/eng_*.log
pids=()
for mc in $MODES; do m=${mc%%:*}; c=${mc##*:}; for p in $(seq 1 $c); do
  python tools/stress_engines.py $m $S > gpurun_out/stress/eng_m${m}_$p.log 2>&1 & pids+=($!)
done; done
for p in "${pids[@]}"; do wait $p; done
grep -h "WRONG\|^mode\|Error" gpurun_out/stress/eng_*.log | sort | cut -c1-300
