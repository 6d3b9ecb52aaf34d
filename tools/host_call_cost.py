"""What a call with HOST arrays costs around its kernel (the reference's calling convention: NumPy in, NumPy
out -- lib.migrate_and_find_max, MigrationScan._compute): wall per Engine.detect(host log-onsets) against the
stacking kernel's own time.  (development aid)   usage: host_call_cost.py CONFIG [reps]"""
import json
import sys
import time
import pathlib

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from quakemigrate_amd import synth  # noqa: E402
from quakemigrate_amd.core import lib  # noqa: E402

cfg = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
case = synth.make_case(cfg)
lon = np.ascontiguousarray(np.log(np.clip(case.onsets, 0.01, np.inf)))
eng = lib.Engine(0)
eng.load_lut(case.traveltimes)
out = (np.zeros(case.n_samples), np.zeros(case.n_samples), np.zeros(case.n_samples, dtype=np.int64))
for _ in range(5):
    eng.detect(lon, case.fsmp, case.lsmp, case.available, out=out)
eng.config("log_timing", 1)
t0 = time.perf_counter()
for _ in range(reps):
    eng.detect(lon, case.fsmp, case.lsmp, case.available, out=out)
wall = (time.perf_counter() - t0) / reps * 1e3
kms, calls = eng.kernel_log()
print(json.dumps({"config": cfg, "host_call_ms": round(wall, 4), "kernel_ms": round(kms / calls, 4),
                  "around_the_kernel_ms": round(wall - kms / calls, 4), "idx_sum": int(out[2].sum())}))
