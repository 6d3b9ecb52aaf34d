import sys, time, json, pathlib
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from quakemigrate_amd import synth
from quakemigrate_amd.core import lib
from quakemigrate_amd.stream import StreamingDetector
case = synth.make_case("C1", step=0)
wins = [np.ascontiguousarray(np.log(np.clip(synth.make_case("C1", step=s, table=False).onsets, 0.01, np.inf))) for s in range(8)]
eng = lib.Engine(0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.load_lut(case.traveltimes)
S, T = wins[0].shape
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
