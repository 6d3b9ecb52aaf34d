#!/bin/bash
# round 5, sixth GPU call: loop knobs (quads fetched unconditionally, stream prefetch) on C3 / C4 / locate,
# the rows SURVEY 8(f) marks next incl. the drop-in migrate's volume rate
cd "$(dirname "$0")/.."
O=gpurun_out/r05_check6; mkdir -p $O
V=build_variants
{
python tools/ab.py --config C3 --steps 6 - $V/libqmhip_nq5.so $V/libqmhip_nq6.so $V/libqmhip_pf0.so $V/libqmhip_pfe2.so -
python tools/ab.py --config C4 --steps 3 --case '{"x_range": [150, 200]}' - $V/libqmhip_nq5.so $V/libqmhip_pf0.so $V/libqmhip_pfe2.so
python tools/ab.py --config C3L --mode volume --steps 8 - $V/libqmhip_nq5.so $V/libqmhip_pf0.so $V/libqmhip_pfe2.so
python tools/ab.py --config C1 --steps 20 - $V/libqmhip_nq5.so $V/libqmhip_pf0.so $V/libqmhip_pfe2.so
} 2>&1 | tee $O/ab.txt
python tools/widen_bench.py > $O/widen_rows.jsonl 2> $O/widen.err; cat $O/widen_rows.jsonl; tail -3 $O/widen.err
