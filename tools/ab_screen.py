#!/usr/bin/env python3
"""A/B of engine builds on the C3 detect step (development aid).
usage: ab_screen.py [--config C3] [--steps 8] lib1.so lib2.so ...   ('-' = the in-tree build)"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys, time
import numpy as np, torch
sys.path.insert(0, %(root)r)
from quakemigrate_amd import synth
from quakemigrate_amd.core import lib
cfg = json.loads(%(cfg)r)
case = synth.make_case(%(config)r, step=0, **json.loads(%(case)r))
eng = lib.Engine(0, **cfg)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.load_lut(case.traveltimes)
lon = torch.from_numpy(np.log(np.clip(case.onsets, 0.01, np.inf))).cuda()
ns = case.n_samples
out = (torch.zeros(ns, dtype=torch.float64, device="cuda"), torch.zeros(ns, dtype=torch.float64, device="cuda"),
       torch.zeros(ns, dtype=torch.int64, device="cuda"))
for _ in range(2):
    eng.detect(lon, case.fsmp, case.lsmp, case.available, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(%(steps)d):
    eng.detect(lon, case.fsmp, case.lsmp, case.available, out=out)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / %(steps)d * 1e3
extra = {}
for k in ("screened_steps", "fallback_steps", "last_candidates"):
    try: extra[k] = eng.get(k)
    except Exception: pass
print(json.dumps({"ms": round(ms, 3), "Gns": round(case.traveltimes[..., 0].size * ns / ms / 1e6, 1), **extra,
                  "idx_sum": int(out[2].sum().item()), "coa_sum": float(out[0].sum().item()),
                  "norm_sum": float(out[1].sum().item())}))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--engine", default='{"screen": 1}')
    ap.add_argument("--case", default="{}", help='make_case kwargs, e.g. {"x_range": [150, 200]}')
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    for lib in args.libs:
        env = dict(os.environ)
        if lib != "-":
            env["QM_HIP_LIB"] = os.path.abspath(lib)
        code = CHILD % dict(root=ROOT, cfg=args.engine, config=args.config, steps=args.steps, case=args.case)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(lib, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)


if __name__ == "__main__":
    main()
