#!/usr/bin/env python3
"""Shift-reuse kernel vs the round-2 kernels and the oracle on small shapes (development aid)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quakemigrate_amd import synth
from quakemigrate_amd.core import lib
from oracle import qm_oracle

bad = 0
cases = [("C2", dict(grid=(26, 25, 14), n_samples=700)), ("C2", dict(grid=(17, 9, 11), n_samples=300, rows=7)),
         ("C3", dict(grid=(40, 33, 21), n_samples=1000)), ("C1", dict(grid=(23, 20, 19), n_samples=625)),
         ("C3", dict(grid=(16, 16, 16), n_samples=256, rows=1)), ("C3", dict(grid=(33, 18, 9), n_samples=513, rows=31)),
         ("C4", dict(grid=(24, 24, 16), n_samples=900, rows=60)), ("C2", dict(grid=(5, 3, 2), n_samples=260, rows=4)),
         ("C4", dict(grid=(33, 26, 18), n_samples=513, rows=47)), ("C4", dict(grid=(20, 21, 22), n_samples=300, rows=64)),
         ("C3", dict(grid=(18, 17, 16), n_samples=400, rows=33))]
for name, kw in cases:
    case = synth.make_case(name, step=0, **kw)
    lon = np.log(np.clip(case.onsets, 0.01, np.inf))
    res = {}
    for tag, cfg in (("shift", {"shift_waves": int(os.environ.get("QM_CHECK_WAVES", "0"))}), ("round2", {"shift": 0})):
        eng = lib.Engine(0, **cfg)
        eng.load_lut(case.traveltimes)
        res[tag] = eng.detect(lon, case.fsmp, case.lsmp, case.available)
        res[tag + "_kernel"] = eng.get("last_kernel")
        if tag == "shift":
            info = (eng.get("shift_ok"), eng.get("shift_brick_nodes"), eng.get("shift_wide_bricks"))
        eng.close()
    ra, rb, rc = qm_oracle.detect(case.onsets, case.traveltimes, case.fsmp, case.lsmp, case.available, threads=8)
    a, b, c = res["shift"]
    a2, b2, c2 = res["round2"]
    ok = (np.array_equal(c, rc) and np.array_equal(c, c2) and np.array_equal(a, a2)
          and np.allclose(a, ra, rtol=1e-12, atol=0) and np.allclose(b, rb, rtol=1e-11, atol=0)
          and np.allclose(b, b2, rtol=1e-12, atol=0))
    print(name, kw, "kernel", res["shift_kernel"], "(shift_ok, brick nodes, wide)", info, "OK" if ok else "MISMATCH",
          "idx!=oracle", int((c != rc).sum()), "idx!=round2", int((c != c2).sum()),
          "max rel", float(np.max(np.abs(a - ra) / ra)), "norm rel", float(np.max(np.abs(b - rb) / rb)), flush=True)
    bad += not ok
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
