#!/usr/bin/env python3
"""Which engine of the fuzz campaign's pair goes wrong under GPU sharing?  (development aid)

One table and one onset set (the shape of a failing trial of tools/fuzz_shift.py), the oracle once, then for
`seconds`: engines made, used once and destroyed, in the order given by `mode`, every result against the oracle.
  mode 0: Engine(automatic) then Engine(shift=0)        (the campaign's order)
  mode 1: Engine(shift=0) then Engine(automatic)
  mode 2: Engine(shift=0) only, a new one every iteration
  mode 3: ONE Engine(shift=0), detect repeated
  mode 4 (round 5): as mode 0, but ALTERNATING TWO DIFFERENT TABLES (other delays, other row count) -- a block
          the pool recycles then holds the OTHER table's derived data, so a kernel that reads scratch it has not
          written cannot find the right values there by accident; run it with QM_HIP_POOL_POISON=1 as well
usage: stress_engines.py mode seconds"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import qm_oracle  # noqa: E402
from quakemigrate_amd.core import lib  # noqa: E402

mode, seconds = int(sys.argv[1]), float(sys.argv[2])
def table(seed, grid, S, ns, fsmp, lsmp):
    rng = np.random.default_rng(seed)
    ijk = np.stack(np.indices(grid), axis=-1).astype(np.float64)
    tt = np.empty(grid + (S,), dtype=np.int32)
    for r in range(S):
        src = rng.uniform(-5, np.array(grid) + 5)
        d = np.sqrt(((ijk - src) ** 2).sum(-1)) * rng.uniform(0.2, 7.0)
        tt[..., r] = np.minimum(np.rint(d - d.min() + rng.integers(0, 5)), lsmp).astype(np.int32)
    lon = np.log(np.clip(rng.lognormal(0, 0.6, size=(S, fsmp + ns + lsmp)), 0.01, None))
    want = qm_oracle.detect(lon, tt, fsmp, lsmp, S, threads=4, prelogged=True)
    return tt, lon, fsmp, lsmp, S, want


tables = [table(7, (9, 29, 20), 157, 753, 11, 120)]
if mode == 4:
    tables.append(table(8, (12, 22, 18), 38, 600, 7, 90))
cfgs = {"auto": dict(shift_rows_direct=1), "round2": dict(shift=0, shift_lazy=-1)}
order = {0: ("auto", "round2"), 1: ("round2", "auto"), 2: ("round2",), 3: ("round2",), 4: ("auto", "round2")}[mode]
bad = {k: 0 for k in cfgs}
runs = {k: 0 for k in cfgs}
kept = None
t0 = time.time()
it = 0
while time.time() - t0 < seconds:
    it += 1
    tt, lon, fsmp, lsmp, avail, want = tables[it % len(tables)]
    for tag in order:
        if mode == 3 and kept is not None:
            eng = kept
        else:
            eng = lib.Engine(0, **cfgs[tag])
            eng.load_lut(tt)
        got = eng.detect(lon, fsmp, lsmp, avail)
        runs[tag] += 1
        ok = np.array_equal(got[2], want[2]) and np.allclose(got[0], want[0], rtol=1e-13, atol=0)
        if not ok:
            bad[tag] += 1
            wrong = np.flatnonzero((got[2] != want[2]) | ~np.isclose(got[0], want[0], rtol=1e-13, atol=0))
            print("WRONG", tag, "iteration", runs[tag], "kernel", eng.get("last_kernel"), "samples", wrong.size, wrong[:12],
                  "zeros", int((got[0][wrong] == 0).sum()), flush=True)
        if mode == 3:
            kept = eng
        else:
            eng.close()
print("mode", mode, "runs", runs, "wrong", bad, flush=True)
