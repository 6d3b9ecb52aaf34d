"""Where does the native streaming pipeline (qm_stream_*) spend its time?  (development aid)

usage: diag_stream.py CONFIG K [depth] [steps] [engine config json, e.g. '{"stream_stamps": 1}']
       (DIAG_WARM = launches before the clock starts, DIAG_SLEEP = seconds of pause before it)
Prints per step: wall with the copies inside, the stacking kernel's own time (HIP events), the host's time in
push / pop, against the resident step."""
import json
import sys
import time
import pathlib

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from quakemigrate_amd import synth  # noqa: E402
from quakemigrate_amd.core import lib  # noqa: E402
from quakemigrate_amd.stream import StreamingDetector  # noqa: E402

cfg, K = sys.argv[1], int(sys.argv[2])
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 400
case = synth.make_case(cfg, step=0)
wins = [np.ascontiguousarray(np.log(np.clip(synth.make_case(cfg, step=s, table=False).onsets, 0.01, np.inf)))
        for s in range(8)]
eng = lib.Engine(0, **(json.loads(sys.argv[5]) if len(sys.argv) > 5 else {}))
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.load_lut(case.traveltimes)
S, T = wins[0].shape
dev = torch.from_numpy(np.stack(wins[:K])).cuda()
out = tuple(torch.empty((K, case.n_samples), dtype=d, device="cuda") for d in (torch.float64, torch.float64, torch.int64))
for _ in range(3):
    eng.detect_batch(dev, case.fsmp, case.lsmp, case.available, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps // K):
    eng.detect_batch(dev, case.fsmp, case.lsmp, case.available, out=out)
torch.cuda.synchronize()
resident = (time.perf_counter() - t0) / (steps // K * K)
sd = StreamingDetector(eng, S, T, case.fsmp, case.lsmp, case.available, depth=depth, steps_per_launch=K)
import os
sd.run(wins[i % 8] for i in range(int(os.environ.get("DIAG_WARM", 4)) * K))
time.sleep(float(os.environ.get("DIAG_SLEEP", 0)))
eng.config("log_timing", 1)
t_push = t_pop = 0.0
n_full = 0
slow = []                                    # host calls of more than 2 ms: (step, call, ms)
t0 = time.perf_counter()
done = 0
for i in range(steps):
    w = wins[i % 8]
    while True:
        a = time.perf_counter()
        ok = sd.push(w)
        t_push += time.perf_counter() - a
        if time.perf_counter() - a > 2e-3:
            slow.append((i, "push", round((time.perf_counter() - a) * 1e3, 2)))
        if ok:
            break
        n_full += 1
        a = time.perf_counter()
        sd.pop(min(K, sd.pending()[0]))
        t_pop += time.perf_counter() - a
        if time.perf_counter() - a > 2e-3:
            slow.append((i, "pop", round((time.perf_counter() - a) * 1e3, 2)))
        done += K
sd.flush()
a = time.perf_counter()
left = sd.pending()[0]
sd.pop(left)
t_pop += time.perf_counter() - a
wall = (time.perf_counter() - t0) / steps
kms, calls = eng.kernel_log()
print(json.dumps({"config": cfg, "K": K, "depth": depth, "steps": steps, "resident_ms": round(resident * 1e3, 4),
                  "with_copies_ms": round(wall * 1e3, 4), "ratio": round(wall / resident, 3),
                  "kernel_ms_per_step": round(kms / steps, 4), "launches": calls,
                  "host_push_ms_per_step": round(t_push / steps * 1e3, 4),
                  "host_pop_ms_per_step": round(t_pop / steps * 1e3, 4), "ring_full": n_full,
                  "slow_host_calls": slow}))
