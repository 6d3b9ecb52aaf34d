#!/bin/bash
# round 5, second GPU call: the suite on the new runtime (two-half bounce, per-device pool, table_select
# order, packed records everywhere), the suite again with QM_HIP_POOL_POISON=1, and two A/Bs: the
# no-staging timing experiment on the C4 slab / C3 (what overlapping the staging could buy at most),
# degree-8 stored values on the locate volume
cd "$(dirname "$0")/.."
O=gpurun_out/r05_check2; mkdir -p $O
V=build_variants
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/suite.txt 2>&1
( time QM_HIP_POOL_POISON=1 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $O/suite_poison.txt 2>&1
{
python tools/ab.py --config C4 --steps 3 --case '{"x_range": [150, 200]}' - $V/libqmhip_nostage.so
python tools/ab.py --config C3 --steps 6 - $V/libqmhip_nostage.so
python tools/ab.py --config C3L --mode volume --steps 8 - $V/libqmhip_deg8.so - $V/libqmhip_deg8.so
} 2>&1 | tee $O/ab.txt
