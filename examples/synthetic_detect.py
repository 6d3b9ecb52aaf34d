# -*- coding: utf-8 -*-
"""
End-to-end detect sweep on synthetic data, GPU-resident from component waveforms to the
coalescence series -- the hot-path side of ``QuakeScan.detect()``
(quakemigrate/signal/scan.py:297-349, 407-470) without obspy:

    component signals --(onset stage, row f2)--> log-onsets (stay on the GPU)
    float64 travel-time grids --(table serving, row f1)--> resident int32 table
    fused migrate + find_max_coa per timestep (the path) --> max_coa, max_coa_n, argmax
    quantise + STEIM2 (row f4) --> <year>_<julday>.scanmseed, as the reference writes it

Run:  python examples/synthetic_detect.py [out_dir]
"""

import datetime as dt
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from quakemigrate_amd import scanmseed as sm  # noqa: E402
from quakemigrate_amd.core import lib  # noqa: E402


def run(out_dir, grid=(24, 20, 12), spacing=0.5, n_stations=6, rate=50, timestep=6.0,
        n_steps=4, seed=3, steps_per_launch=1):
    """``steps_per_launch`` > 1: the onset rows of that many timesteps are stacked by ONE launch
    (``Engine.detect_batch``) -- what fills the GPU on grids of this size; same series."""
    import torch

    rng = np.random.default_rng(seed)
    nx, ny, nz = grid
    # ---- travel-time grids in seconds, one per station/phase (what a LUT holds) ---------
    st = np.c_[rng.uniform(0, (nx - 1) * spacing, n_stations),
               rng.uniform(0, (ny - 1) * spacing, n_stations), np.zeros(n_stations)]
    gx, gy, gz = np.meshgrid(np.arange(nx) * spacing, np.arange(ny) * spacing,
                             np.arange(nz) * spacing, indexing="ij")
    dist = [np.sqrt((gx - s[0]) ** 2 + (gy - s[1]) ** 2 + (gz - s[2]) ** 2) for s in st]
    grids = [d / 5.0 for d in dist] + [d / 2.9 for d in dist]          # P rows, then S rows
    eng = lib.Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_traveltime_grids(grids)
    eng.serve(rate, np.arange(2 * n_stations))                          # all available
    lut_max = eng.lut_max
    # ---- windowing as QuakeScan does: pre/post pad around every timestep -------------------
    nsta, nlta = int(0.2 * rate) + 1, int(1.0 * rate) + 1
    fsmp = nlta + 20
    lsmp = lut_max + nsta + 20
    ns = int(timestep * rate)
    T = fsmp + ns + lsmp
    total = fsmp + n_steps * ns + lsmp
    # ---- synthetic component waveforms: noise + one event per timestep ---------------------
    n_rows = 2 * n_stations
    trace_row = np.repeat(np.arange(n_rows), 2).astype(np.int32)        # 2 components per row
    wave = rng.standard_normal((len(trace_row), total))
    truth = []
    for k in range(n_steps):
        node = tuple(int(rng.integers(2, d - 2)) for d in grid)
        t0 = fsmp + k * ns + int(rng.integers(ns // 4, 3 * ns // 4))
        truth.append((node, t0 - fsmp))
        for row in range(n_rows):
            arrival = t0 + int(np.rint(grids[row][node] * rate))
            wave[2 * row:2 * row + 2, arrival:arrival + nsta] *= 9.0
    K = max(1, int(steps_per_launch))
    d_log = torch.empty((K, n_rows, T), dtype=torch.float64, device="cuda")
    out = tuple(torch.empty((K, ns), dtype=d, device="cuda")
                for d in (torch.float64, torch.float64, torch.int64))
    series = {k: [] for k in ("coa", "coa_n", "idx")}
    for k0 in range(0, n_steps, K):
        n = min(K, n_steps - k0)
        for j in range(n):                                  # the onset stage, timestep by timestep
            window = np.ascontiguousarray(wave[:, (k0 + j) * ns: (k0 + j) * ns + T])
            eng.onsets(window, trace_row, [nsta] * n_rows, [nlta] * n_rows, transform="energy",
                       position="classic", taper_pad=-1, min_onset_value=0.4, log_out=d_log[j])
        if K == 1:
            eng.detect(d_log[0], fsmp, lsmp, n_rows, out=tuple(o[0] for o in out))
        else:                                               # ... the migration of n timesteps at once
            eng.detect_batch(d_log[:n], fsmp, lsmp, n_rows, out=tuple(o[:n] for o in out))
        series["coa"].append(out[0][:n].reshape(-1).cpu().numpy())
        series["coa_n"].append(out[1][:n].reshape(-1).cpu().numpy())
        series["idx"].append(out[2][:n].reshape(-1).cpu().numpy())
    coa, coa_n, idx = (np.concatenate(series[k]) for k in ("coa", "coa_n", "idx"))
    coord = np.stack(np.unravel_index(idx, grid), axis=-1) * spacing    # index2coord, no pyproj
    # ---- write what detect() writes ------------------------------------------------------------
    out_dir = pathlib.Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    t_start = dt.datetime(2024, 2, 29, 12, 0, 0)
    path = out_dir / f"{t_start.year}_{t_start.timetuple().tm_yday:03d}.scanmseed"
    sm.write_scanmseed(path, t_start, float(rate), sm.quantise(coa, coa_n, coord, ucf=1000.0))
    eng.close()
    return dict(path=path, coa=coa, coa_n=coa_n, idx=idx, coord=coord, truth=truth, grid=grid,
                n_samples=ns)


if __name__ == "__main__":
    res = run(sys.argv[1] if len(sys.argv) > 1 else "synthetic_detect_out")
    for node, t in res["truth"]:
        k = int(np.argmax(res["coa"][max(0, t - 10): t + 10])) + max(0, t - 10)
        print(f"event at node {node}, sample {t}: peak coalescence {res['coa'][k]:.2f} at sample "
              f"{k}, node {np.unravel_index(res['idx'][k], res['grid'])}")
    print("wrote", res["path"])
