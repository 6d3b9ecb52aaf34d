#!/bin/bash
# (1) one process alone, a new stream every iteration; (2) 16 x 120 short processes, one stream each
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r04_d2h; rm -f gpurun_out/r04_d2h/*.log
echo "== alone: $(./tools/micro/d2h_order 4 45 2000 1 1000000000 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' | cut -c1-400)"
pids=()
for p in $(seq 1 16); do
  ( for i in $(seq 1 120); do ./tools/micro/d2h_order 4 0.25 2000 1000000000 1000000000 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r04_d2h/short_$p.log ) & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
echo "== short processes: $(cat gpurun_out/r04_d2h/short_*.log | grep -c iterations) runs, $(cat gpurun_out/r04_d2h/short_*.log | awk '/iterations,/ {for (i=1;i<=NF;i++) if ($i=="iterations,") {it+=$(i-1); bad+=$(i+1)}} END {print it " iterations, " bad " wrong"}')"
grep -h "elements wrong" gpurun_out/r04_d2h/short_*.log | head -5
