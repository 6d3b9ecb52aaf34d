#!/bin/bash
# round 5, third GPU call: the new tests; where the 8-wave shape loses (rows sweep on the C3 grid, both
# shapes at 30 rows; stream-prefetch distance and the no-scalar-load bound on the C4 slab); engines over two
# alternating tables in 16 processes with poisoned pool blocks; step_with_copies through the native pipeline
cd "$(dirname "$0")/.."
O=gpurun_out/r05_check3; mkdir -p $O
V=build_variants
( python -m pytest tests -m gpu -x -q -k "foreign or alternating or native_stream or batch or streaming or c5_" 2>&1 | tail -5 ) > $O/new_tests.txt 2>&1
{
for R in 16 30 40; do python tools/ab.py --config C3 --steps 4 --case "{\"rows\": $R, \"n_samples\": 1536}" --engines '[{}, {"shift_waves": 8}]' - ; done
for R in 44 60; do python tools/ab.py --config C3 --steps 4 --case "{\"rows\": $R, \"n_samples\": 1536}" - ; done
python tools/ab.py --config C4 --steps 3 --case '{"x_range": [150, 200]}' - $V/libqmhip_pf32.so $V/libqmhip_pf64.so $V/libqmhip_nosmem.so
} 2>&1 | tee $O/ab.txt
S=45
mkdir -p $O/stress; pids=()
for p in $(seq 1 16); do QM_HIP_POOL_POISON=1 python tools/stress_engines.py 4 $S > $O/stress/eng_$p.log 2>&1 & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
grep -h "WRONG\|^mode\|Error" $O/stress/eng_*.log | sort | cut -c1-200 > $O/stress.txt
for cfg in C1 E1 E2; do
  python bench.py --config $cfg --steps 400 --warmup 8 --steps-per-launch 8 --no-cpu-baseline --no-screened --no-materialised --no-table-switch > $O/bench_${cfg}_k8.json 2>$O/bench_${cfg}_k8.err
  python bench.py --config $cfg --steps 400 --warmup 8 --steps-per-launch 1 --no-cpu-baseline --no-screened --no-materialised --no-table-switch > $O/bench_${cfg}_k1.json 2>$O/bench_${cfg}_k1.err
done
python bench.py --config C3 --steps 20 --warmup 3 --no-cpu-baseline --no-screened --no-materialised --no-table-switch > $O/bench_C3.json 2>$O/bench_C3.err
python - <<'PY' > $O/copies.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_check3/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d.get("step_with_copies", {})
        print(f.split("/")[-1], "resident ms", round(d["ms_per_step"], 4), "with copies", round(c.get("ms_per_step", float("nan")), 4),
              "ratio", round(c.get("ms_per_step", float("nan")) / d["ms_per_step"], 3), "identical", c.get("identical_to_resident_run"))
    except Exception as e:
        print(f, "ERR", e)
PY
