#!/bin/bash
# 16 concurrent processes for S seconds (tools/micro/d2h_order.hip): three pageable copies behind two kernels,
# (a) as the engine did, stream and buffers new every iteration; (b) the same with the wait the engine now does;
# (c) only the stream new; (d) only the buffers new
cd ${GRAFT_REPO_ROOT:-$(pwd)}
S=${1:-60}
mkdir -p gpurun_out/r04_d2h; rm -f gpurun_out/r04_d2h/*.log
pids=()
for p in 1 2 3 4; do
  ./tools/micro/d2h_order 4 $S 2000 1 1 > gpurun_out/r04_d2h/a_$p.log 2>&1 & pids+=($!)
  ./tools/micro/d2h_order 5 $S 2000 1 1 > gpurun_out/r04_d2h/b_$p.log 2>&1 & pids+=($!)
  ./tools/micro/d2h_order 4 $S 2000 1 1000000000 > gpurun_out/r04_d2h/c_$p.log 2>&1 & pids+=($!)
  ./tools/micro/d2h_order 4 $S 2000 1000000000 1 > gpurun_out/r04_d2h/d_$p.log 2>&1 & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
for f in gpurun_out/r04_d2h/*.log; do echo "$(basename $f): $(grep -v amdgpu.ids $f | tail -3 | tr "\n" " " | cut -c1-330)"; done
