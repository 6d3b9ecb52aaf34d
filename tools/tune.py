# -*- coding: utf-8 -*-
"""GPU-side sweep of engine tunables on the bench workloads (development aid)."""

import argparse
import itertools
import json
import sys
import time
import pathlib

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from quakemigrate_amd import synth  # noqa: E402
from quakemigrate_amd.core import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C2")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--sweep", default="default")
    ap.add_argument("--ns", type=int, default=0, help="override n_samples")
    ap.add_argument("--rows", type=int, default=0, help="override the number of table rows")
    ap.add_argument("--x-range", default="", help="x-planes lo,hi of the grid (one rank's slab of a sharded config)")
    ap.add_argument("--volume", action="store_true", help="materialising path (device volume)")
    ap.add_argument("--marginal", action="store_true",
                    help="the marginalised map of the central half of the scan instead of the volume")
    args = ap.parse_args()
    t0 = time.time()
    xr = tuple(int(v) for v in args.x_range.split(",")) if args.x_range else None
    case = synth.make_case(args.config, n_samples=args.ns or None, rows=args.rows or None, x_range=xr)
    vol = None
    cmap = None
    if args.marginal:
        import torch
        cmap = torch.empty(case.n_nodes_total, dtype=torch.float64, device="cuda")
        scan = tuple(torch.empty(case.n_samples, dtype=d, device="cuda")
                     for d in (torch.float64, torch.float64, torch.int64))
        lon_dev = torch.from_numpy(np.ascontiguousarray(np.log(np.clip(case.onsets, 0.01, np.inf)))).cuda()
    if args.volume:
        import torch
        vol = torch.empty((case.n_nodes_total, case.n_samples), dtype=torch.float64, device="cuda")
        scan = tuple(torch.empty(case.n_samples, dtype=d, device="cuda")
                     for d in (torch.float64, torch.float64, torch.int64))
    print(f"case {args.config} built in {time.time()-t0:.1f}s grid={case.grid} S={case.available} "
          f"ns={case.n_samples} lutmax={case.traveltimes.max()}", flush=True)
    lon = np.ascontiguousarray(np.log(np.clip(case.onsets, 0.01, np.inf)))
    work = case.n_nodes_total * case.n_samples
    if args.sweep == "default":
        grid = [dict(samples_per_lane=j, waves=w, lds_bytes=l, brick=b)
                for j, w, l, b in itertools.product(
                    [4, 2], [8, 16], [80 * 1024, 160 * 1024, 53 * 1024],
                    [(4, 4, 8), (4, 4, 4), (2, 4, 8), (8, 8, 8)])]
    else:
        grid = json.loads(args.sweep)
    ref = None
    for cfg in grid:
        cfg = dict(cfg)
        bx, by, bz = cfg.pop("brick", (0, 0, 0))
        try:
            eng = lib.Engine(0, brick_x=bx, brick_y=by, brick_z=bz, **cfg)
            eng.load_lut(case.traveltimes)
            wide = eng.get("n_wide_bricks")
            best = 1e9
            for _ in range(args.reps):
                if cmap is not None:
                    ns = case.n_samples
                    eng.marginal_map(lon_dev, case.fsmp, case.lsmp, case.available, ns // 4, ns - ns // 4,
                                     out=cmap, scan_out=scan)
                    out = tuple(t.cpu().numpy() for t in scan)
                elif vol is None:
                    out = eng.detect(lon, case.fsmp, case.lsmp, case.available)
                else:
                    eng.migrate(lon, case.fsmp, case.lsmp, case.available, vol, scan_out=scan)
                    out = tuple(t.cpu().numpy() for t in scan)
                best = min(best, eng.last_kernel_ms())
            if ref is None:
                ref = out
            same = bool(np.array_equal(out[2], ref[2]) and np.allclose(out[0], ref[0], rtol=1e-12))
            bx, by, bz = eng.get("brick_x"), eng.get("brick_y"), eng.get("brick_z")
            print(json.dumps(dict(cfg=cfg, brick=[bx, by, bz], wide=wide, nbricks=eng.get("n_bricks"),
                                  ms=round(best, 3), gns=round(work / best / 1e6, 2), same=same)),
                  flush=True)
            eng.close()
        except Exception as e:  # noqa: BLE001
            print(json.dumps(dict(cfg=cfg, brick=[bx, by, bz], error=str(e)[:200])), flush=True)


if __name__ == "__main__":
    main()
