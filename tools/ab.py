#!/usr/bin/env python3
"""A/B of engine builds and engine configurations on one workload (development aid).

usage: ab.py [--config C3] [--mode detect|volume|marginal|scan] [--steps 6]
             [--engines '[{"exact": 0}, {"exact": 1}]'] lib1.so lib2.so ...   ('-' = in-tree build)

Every (library, engine configuration) pair runs in its own process; the line printed carries the
step time and checksums of the outputs so that variants can be compared bit for bit."""
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
mode = %(mode)r
case = synth.make_case(%(config)r, step=0, **json.loads(%(case)r))
eng = lib.Engine(0, **cfg)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.load_lut(case.traveltimes)
lon = torch.from_numpy(np.log(np.clip(case.onsets, 0.01, np.inf))).cuda()
ns = case.n_samples
n = int(np.prod(case.traveltimes.shape[:3]))
out = (torch.zeros(ns, dtype=torch.float64, device="cuda"), torch.zeros(ns, dtype=torch.float64, device="cuda"),
       torch.zeros(ns, dtype=torch.int64, device="cuda"))
vol = torch.zeros((n, ns), dtype=torch.float64, device="cuda") if mode in ("volume", "scan") else None
if mode == "scan":            # find_max_coa of a resident volume, written once
    eng.migrate(lon, case.fsmp, case.lsmp, case.available, vol)
cmap = torch.zeros(n, dtype=torch.float64, device="cuda") if mode == "marginal" else None
def step():
    if mode == "detect":
        eng.detect(lon, case.fsmp, case.lsmp, case.available, out=out)
    elif mode == "volume":
        eng.migrate(lon, case.fsmp, case.lsmp, case.available, vol, scan_out=out)
    elif mode == "scan":
        eng.find_max_coa(vol, ns, n, out)
    else:
        eng.marginal_map(lon, case.fsmp, case.lsmp, case.available, ns // 4, ns - ns // 4, out=cmap, scan_out=out)
for _ in range(2):
    step()
torch.cuda.synchronize()
eng.config("log_timing", 1)
t0 = time.perf_counter()
for _ in range(%(steps)d):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / %(steps)d * 1e3
kms, calls = eng.kernel_log()
extra = {}
if vol is not None:
    extra["vol_sum"] = float(vol.sum().item())
if cmap is not None:
    extra["map_sum"] = float(cmap.sum().item())
print(json.dumps({"ms": round(ms, 3), "kernel_ms": round(kms / max(calls, 1), 3),
                  "Gns": round(n * ns / ms / 1e6, 1),
                  "idx_sum": int(out[2].sum().item()), "coa_sum": float(out[0].sum().item()),
                  "norm_sum": float(out[1].sum().item()), **extra}))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--mode", default="detect", choices=["detect", "volume", "marginal", "scan"])
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--engines", default="[{}]", help="json list of engine configurations")
    ap.add_argument("--case", default="{}", help='make_case kwargs, e.g. {"x_range": [150, 200]}')
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    for lib in args.libs:
        for cfg in json.loads(args.engines):
            env = dict(os.environ)
            if lib != "-":
                env["QM_HIP_LIB"] = os.path.abspath(lib)
            code = CHILD % dict(root=ROOT, cfg=json.dumps(cfg), config=args.config, steps=args.steps,
                                case=args.case, mode=args.mode)
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
            last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:]
            print(os.path.basename(lib), args.config, args.mode, json.dumps(cfg), last, flush=True)


if __name__ == "__main__":
    main()
