"""Where does StreamingDetector spend its time on short steps?  (development aid)"""
import cProfile
import pstats
import sys
import time
import pathlib

import numpy as np
import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from quakemigrate_amd import synth  # noqa: E402
from quakemigrate_amd.core import lib  # noqa: E402
from quakemigrate_amd.stream import StreamingDetector  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
case = synth.make_case(cfg, step=0)
host = np.ascontiguousarray(np.log(np.clip(case.onsets, 0.01, np.inf)))
eng = lib.Engine(0)
eng.load_lut(case.traveltimes)
sd = StreamingDetector(eng, case.available, host.shape[1], case.fsmp, case.lsmp, case.available, depth=3)
sd.run(host for _ in range(3))
torch.cuda.synchronize()
for steps in (30,):
    t0 = time.perf_counter()
    sd.run(host for _ in range(steps))
    torch.cuda.synchronize()
    print(cfg, "streaming ms/step", (time.perf_counter() - t0) / steps * 1e3)
pr = cProfile.Profile()
pr.enable()
sd.run(host for _ in range(30))
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
# resident-input loop for comparison
eng.set_stream(torch.cuda.current_stream().cuda_stream)
d = torch.from_numpy(host).cuda()
out = tuple(torch.empty(case.n_samples, dtype=t, device="cuda") for t in (torch.float64, torch.float64, torch.int64))
eng.detect(d, case.fsmp, case.lsmp, case.available, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    eng.detect(d, case.fsmp, case.lsmp, case.available, out=out)
torch.cuda.synchronize()
print(cfg, "resident ms/step", (time.perf_counter() - t0) / 30 * 1e3)
