# -*- coding: utf-8 -*-
"""
Timing of the rows SURVEY 8(f) marks "next", at the C3 geometry, each beside the NumPy / SciPy
restatement of what the reference does on the host (oracle/qm_oracle.py; test infrastructure).

  f1  table serving   : float64 grids [S][N] -> int32 table [N][S], rint(tt * rate)   (lut.py:502-538)
  f2  onset stage     : STA/LTA + RMS + clip + log of 3-component traces              (stalta.py:491-583)
  f3  location fits   : normalise, 2-pass Gaussian smoothing, covariance moments, windows (scan.py:696-1077)

usage (GPU box): python tools/widen_bench.py [--no-cpu] > out.jsonl
"""
import argparse
import json
import pathlib
import sys
import time

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from quakemigrate_amd import locate, synth  # noqa: E402
from quakemigrate_amd.core import lib  # noqa: E402


def gpu_ms(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    from oracle import qm_oracle as oq

    cfg = synth.CONFIGS["C3"]
    grid, S, rate = cfg["grid"], cfg["rows"], cfg["rate"]
    n = int(np.prod(grid))
    eng = lib.Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)

    # ---- f1: table serving ---------------------------------------------------------------
    case = synth.make_case("C3", step=0, n_samples=64)
    tt_s = (np.maximum(case.traveltimes, 0).astype(np.float64) + 0.25) / rate     # seconds, off-grid
    grids = [np.ascontiguousarray(tt_s[..., r]) for r in range(S)]
    eng.set_traveltime_grids(grids)
    rows = list(range(S))
    ms = gpu_ms(lambda: eng.serve(rate, rows), 5)
    bytes_alg = 8.0 * n * S + 4.0 * n * S
    # the two halves: the serving kernel itself (8 B read + 4 B written per table entry) and what
    # qm_engine_load_lut does with ANY new table (device copy, brick records, window offsets)
    served = torch.from_numpy(eng.download_lut()).cuda()
    ms_prep = gpu_ms(lambda: eng.load_lut(served), 5)
    ms_kernel = max(ms - ms_prep, 1e-3)
    out = {"row": "f1", "what": f"serve {S} float64 grids of {grid} -> int32 table, incl. the engine's "
                                 "brick-table preparation for the new table",
           "gpu_ms": round(ms, 3), "algorithmic_GB": round(bytes_alg / 1e9, 3),
           "serve_kernel_ms": round(ms_kernel, 3),
           "serve_kernel_TBps": round(bytes_alg / ms_kernel / 1e9, 3),
           "load_lut_of_a_device_table_ms": round(ms_prep, 3)}
    del served
    if not args.no_cpu:
        t0 = time.perf_counter()
        want = oq.np_serve_traveltimes(grids, rate)
        out["cpu_numpy_s"] = round(time.perf_counter() - t0, 3)
        out["identical"] = bool(np.array_equal(eng.download_lut().reshape(want.shape), want))
    print(json.dumps(out), flush=True)

    # ---- f2: onset stage -------------------------------------------------------------------
    T = case.fsmp + 6000 + case.lsmp
    rng = np.random.default_rng(5)
    trace_row = np.repeat(np.arange(S, dtype=np.int32), 3)
    sig = rng.standard_normal((len(trace_row), T))
    nsta = np.full(S, 10, dtype=np.int32)
    nlta = np.full(S, 50, dtype=np.int32)
    d_sig = torch.from_numpy(sig).cuda()
    d_log = torch.empty((S, T), dtype=torch.float64, device="cuda")
    ms = gpu_ms(lambda: eng.onsets(d_sig, trace_row, nsta, nlta, taper_pad=10, log_out=d_log), 20)
    out = {"row": "f2", "what": f"onset stage: {len(trace_row)} traces x {T} samples -> {S} log-onset rows "
                                 "(device in, device out)", "gpu_ms": round(ms, 4)}
    if not args.no_cpu:
        t0 = time.perf_counter()
        raw, logged = oq.np_onset_stage(sig, trace_row, nsta, nlta, "energy", "classic", 10, 0.4)
        out["cpu_c_port_s"] = round(time.perf_counter() - t0, 4)
        out["max_rel_diff"] = float(np.max(np.abs(d_log.cpu().numpy() - logged) /
                                           np.maximum(np.abs(logged), 1e-3)))
    print(json.dumps(out), flush=True)

    # ---- f3: location fits -----------------------------------------------------------------
    idx = np.meshgrid(*[np.arange(k) for k in grid], indexing="ij")
    r2 = sum(((g - c) / w) ** 2 for g, c, w in zip(idx, (90.3, 120.6, 40.2), (6.0, 7.0, 5.0)))
    coa = 1.2 + 2.5 * np.exp(-0.5 * r2) + 0.02 * rng.standard_normal(grid)
    d_map = torch.from_numpy(coa).cuda()
    spacing = np.array([0.5, 0.5, 0.5])
    ms = gpu_ms(lambda: locate.calculate_location(eng, d_map, spacing), 10)
    ms_dev = gpu_ms(lambda: eng.locate_fits(d_map, spacing), 10)
    out = {"row": "f3", "what": f"_calculate_location on a {grid} map (device sweeps + host window algebra)",
           "gpu_ms_total": round(ms, 3), "gpu_ms_device_part": round(ms_dev, 3),
           "map_MB": round(8.0 * n / 1e6, 1)}
    if not args.no_cpu:
        t0 = time.perf_counter()
        norm = coa / np.nanmax(coa)
        smoothed = oq.np_gaufilt3d(norm)
        t1 = time.perf_counter()
        oq.np_gaufit3d(smoothed)
        oq.np_covfit3d(norm, spacing)
        oq.np_splineloc(norm)
        out["cpu_numpy_scipy_s"] = round(time.perf_counter() - t0, 3)
        out["cpu_gaufilt3d_s"] = round(t1 - t0, 3)
        fits = locate.calculate_location(eng, d_map, spacing)
        loc, _, _ = oq.np_gaufit3d(smoothed)
        out["gaussian_location_max_abs_diff_nodes"] = float(np.max(np.abs(fits.gaussian - loc)))
    print(json.dumps(out), flush=True)

    # ---- the drop-in symbol: migrate() with the reference's signature at the locate window -----
    # (host arrays in and out: lib.py:99-123 -- the volume, 8 B per node-sample, comes back through the
    # pinned two-half bounce buffer, DMA overlapped with the CPU copy; the table is resident after the
    # first call, the volume is known to be zero)
    import os

    case = synth.make_case("C3L")
    ns = case.n_samples
    os.environ["QM_HIP_GRID"] = ",".join(str(g) for g in grid)
    os.environ["QM_HIP_ASSUME_ZERO_MAP"] = "1"
    lib.migrate(case.onsets, case.traveltimes, case.fsmp, case.lsmp, case.available)       # table upload, sizing
    t0 = time.perf_counter()
    m = lib.migrate(case.onsets, case.traveltimes, case.fsmp, case.lsmp, case.available)
    dt = time.perf_counter() - t0
    gb = 8.0 * n * ns / 1e9
    t0 = time.perf_counter()
    host = np.empty((n, ns))
    host[...] = 0.0
    t_alloc = time.perf_counter() - t0
    out = {"row": "drop-in migrate", "what": f"lib.migrate (reference signature, host arrays) at the C3 locate window: "
                                            f"{grid} x {ns} samples, {gb:.1f} GB volume to the host",
           "wall_s": round(dt, 3), "volume_GBps_wall": round(gb / dt, 1),
           "zero_filling_a_host_volume_of_that_size_s": round(t_alloc, 3),
           "checksum": float(m[::7, ::5, ::3].sum())}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
