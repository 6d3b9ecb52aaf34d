#!/bin/bash
# stress the two trials the 4xx fuzz campaign failed once: the trial's engine sequence R times in each of 16
# concurrent processes (the campaign's own concurrency)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
R=${1:-400}
mkdir -p gpurun_out/r04_stress
pids=()
for p in 1 2 3 4 5 6 7 8; do
  QM_FUZZ_ONLY=52 QM_FUZZ_REPEAT=$R timeout 600 python tools/fuzz_shift.py 53 402 > gpurun_out/r04_stress/t52_$p.log 2>&1 & pids+=($!)
  QM_FUZZ_ONLY=509 QM_FUZZ_REPEAT=$R timeout 600 python tools/fuzz_shift.py 510 407 > gpurun_out/r04_stress/t509_$p.log 2>&1 & pids+=($!)
done
t0=$(date +%s)
bad=0; for p in "${pids[@]}"; do wait $p || bad=$((bad+1)); done; echo "processes failed: $bad"
echo "elapsed $(( $(date +%s) - t0 )) s"
grep -l "MISMATCH\|Error" gpurun_out/r04_stress/*.log
grep -h "MISMATCH" -A8 gpurun_out/r04_stress/*.log | cut -c1-1200 | head -80
tail -qn1 gpurun_out/r04_stress/*.log | cut -c1-200 | sort | uniq -c
