"""Does the one-off stall of a stream pipeline (tools/diag_stream.py, "stream_stamps") come back?  Three streams one
after the other in ONE process (C1, one timestep per launch, depth 3), then a loop of detect + device synchronisation
per step: per loop the wall time per step and every blocking call of more than 2 ms.  (development aid)
usage: diag_stall.py [streams | sync | torch]   (sync / torch: that loop FIRST in the process, then the streams)"""
import sys, time, json, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from quakemigrate_amd import synth
from quakemigrate_amd.core import lib
from quakemigrate_amd.stream import StreamingDetector
case = synth.make_case("C1", step=0)
wins = [np.ascontiguousarray(np.log(np.clip(synth.make_case("C1", step=s, table=False).onsets, 0.01, np.inf))) for s in range(8)]
eng = lib.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.load_lut(case.traveltimes)
S, T = wins[0].shape
mode = sys.argv[1] if len(sys.argv) > 1 else "streams"
if mode == "torch":                 # nothing of this library: a torch kernel + synchronize per step, first thing in the process
    x = torch.randn(4096, 4096, device="cuda", dtype=torch.float32)
    slow = []
    t0 = time.perf_counter()
    for i in range(600):
        y = x @ x                   # ~0.3 ms
        a = time.perf_counter()
        torch.cuda.synchronize()
        if time.perf_counter() - a > 2e-3:
            slow.append((i, round((time.perf_counter() - a) * 1e3, 1)))
    print("torch matmul + synchronize per step: ms/step", round((time.perf_counter() - t0) / 600 * 1e3, 4), "slow syncs", slow, flush=True)
if mode == "sync":                  # detect + device synchronisation per step FIRST (no stream before it)
    dev = torch.from_numpy(wins[0]).cuda()
    out = tuple(torch.empty(case.n_samples, dtype=d, device="cuda") for d in (torch.float64, torch.float64, torch.int64))
    slow = []
    t0 = time.perf_counter()
    for i in range(600):
        eng.detect(dev, case.fsmp, case.lsmp, case.available, out=out)
        a = time.perf_counter()
        torch.cuda.synchronize()
        if time.perf_counter() - a > 2e-3:
            slow.append((i, round((time.perf_counter() - a) * 1e3, 1)))
    print("detect + sync per step, first thing in the process: ms/step", round((time.perf_counter() - t0) / 600 * 1e3, 4), "slow syncs", slow, flush=True)
for rep in range(3):
    sd = StreamingDetector(eng, S, T, case.fsmp, case.lsmp, case.available, depth=3, steps_per_launch=1)
    slow = []
    t0 = time.perf_counter()
    for i in range(500):
        while True:
            if sd.push(wins[i % 8]):
                break
            a = time.perf_counter()
            sd.pop(1)
            if time.perf_counter() - a > 2e-3:
                slow.append((i, round((time.perf_counter() - a) * 1e3, 1)))
    sd.flush(); sd.pop(sd.pending()[0])
    print("stream", rep, "ms/step", round((time.perf_counter() - t0) / 500 * 1e3, 4), "slow pops", slow, flush=True)
    sd.close()
# resident loop with a blocking sync every step
dev = torch.from_numpy(wins[0]).cuda()
out = tuple(torch.empty(case.n_samples, dtype=d, device="cuda") for d in (torch.float64, torch.float64, torch.int64))
slow = []
t0 = time.perf_counter()
for i in range(500):
    eng.detect(dev, case.fsmp, case.lsmp, case.available, out=out)
    a = time.perf_counter()
    torch.cuda.synchronize()
    if time.perf_counter() - a > 2e-3:
        slow.append((i, round((time.perf_counter() - a) * 1e3, 1)))
print("detect + sync each step: ms/step", round((time.perf_counter() - t0) / 500 * 1e3, 4), "slow syncs", slow)
