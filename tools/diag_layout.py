"""What layout the engine picks for a table (development aid): brick shape, samples per lane,
bricks left to the direct kernel, and the detect step time."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from quakemigrate_amd import synth  # noqa: E402
from quakemigrate_amd.core import lib  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "C2"
engine_cfg = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
for rows in [int(v) for v in sys.argv[3:]] or [64, 65, 96, 128]:
    extra = {"x_range": (150, 200)} if config == "C4" else {}
    extra.update(json.loads(os.environ.get("QM_DIAG_CASE", "{}")))
    case = synth.make_case(config, step=0, rows=rows, **extra)
    eng = lib.Engine(0, **engine_cfg)
    eng.load_lut(case.traveltimes)
    lon = torch.from_numpy(np.log(np.clip(case.onsets, 0.01, np.inf))).cuda()
    out = (torch.zeros(case.n_samples, dtype=torch.float64, device="cuda"),
           torch.zeros(case.n_samples, dtype=torch.float64, device="cuda"),
           torch.zeros(case.n_samples, dtype=torch.int64, device="cuda"))
    for _ in range(2):
        eng.detect(lon, case.fsmp, case.lsmp, case.available, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        eng.detect(lon, case.fsmp, case.lsmp, case.available, out=out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    n = int(np.prod(case.traveltimes.shape[:3]))
    info = {k: eng.get(k) for k in ("brick_x", "brick_y", "brick_z", "samples_per_lane",
                                    "n_bricks", "n_wide_bricks", "last_kernel", "last_kernel_j",
                                    "mean_span", "waves")}
    info.update(cfg=engine_cfg, rows=rows, ms=round(ms, 2), Tadds_per_s=round(n * case.n_samples * rows / ms / 1e9, 2),
                lut_max=eng.lut_max)
    print(json.dumps(info))
    eng.close()
