# -*- coding: utf-8 -*-
"""
CPU ORACLE -- test infrastructure, NOT the product path.

Two independent CPU statements of the reference's migrate / find_max_coa path:

* ``c_*``  : ctypes calls into ``oracle/libqm_oracle.so`` (the C restatement in
  ``oracle/qm_oracle.c``, built with the reference's flags);
* ``np_*`` : a NumPy restatement, vectorised over nodes, float64, rows added in
  ascending order (the order that fixes the rounding of the sums).

and, when ``oracle/_ref/qmlib.so`` exists (a build of the REFERENCE's own C
files, see ``oracle/Makefile``), ``ref_*`` wrappers that call the real thing
through the same Python-level semantics as ``quakemigrate/core/lib.py:52-170``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this module.  Parity status: pinned by ``tests/golden`` (generated
from the reference by ``oracle/make_golden.py``).
"""

from __future__ import annotations

import ctypes
import pathlib

import numpy as np
import numpy.ctypeslib as clib

_HERE = pathlib.Path(__file__).resolve().parent

_dp = clib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = clib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = clib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_I32 = ctypes.c_int32
_I64 = ctypes.c_int64

stalta_header_t = np.dtype(
    [("n", np.int32), ("nsta", np.int32), ("nlta", np.int32)], align=True
)
_hdrp = clib.ndpointer(stalta_header_t, flags="C_CONTIGUOUS")

_STACK_ARGS = [_dp, _i32p, _dp, _I32, _I32, _I32, _I32, _I32, _I64, _I64]
_SCAN_ARGS = [_dp, _dp, _dp, _i64p, _I32, _I64, _I64]


def _load(path, names):
    lib = ctypes.CDLL(str(path))
    stack, scan, s_over, s_cent, s_rec = (getattr(lib, n) for n in names)
    stack.argtypes, stack.restype = _STACK_ARGS, None
    scan.argtypes, scan.restype = _SCAN_ARGS, None
    for f in (s_over, s_cent, s_rec):
        f.argtypes, f.restype = [_dp, _hdrp, _dp], None
    return dict(stack=stack, scan=scan, overlapping=s_over, centred=s_cent,
                recursive=s_rec)


_cache = {}


def _port():
    if "port" not in _cache:
        path = _HERE / "libqm_oracle.so"
        if not path.exists():
            raise ImportError(
                f"{path} missing: run `make -C oracle` (or __graft_entry__.build())"
            )
        _cache["port"] = _load(
            path, ["oq_stack", "oq_scan_max", "oq_stalta_overlapping",
                   "oq_stalta_centred", "oq_stalta_recursive"])
    return _cache["port"]


def have_ref() -> bool:
    return (_HERE / "_ref" / "qmlib.so").exists()


def _ref():
    if "ref" not in _cache:
        _cache["ref"] = _load(
            _HERE / "_ref" / "qmlib.so",
            ["migrate", "find_max_coa", "overlapping_sta_lta",
             "centred_sta_lta", "recursive_sta_lta"])
    return _cache["ref"]


# --------------------------------------------------------------------------
# Python-level semantics of quakemigrate/core/lib.py:52-125 and :131-170
# --------------------------------------------------------------------------
def log_onsets(onsets: np.ndarray) -> np.ndarray:
    """clip at 0.01 then natural log (lib.py:93-94)."""
    return np.log(np.clip(onsets, 0.01, np.inf))


def _migrate_with(fns, onsets, traveltimes, first_idx, last_idx, available,
                  threads, prelogged=False):
    lon = np.ascontiguousarray(onsets if prelogged else log_onsets(onsets))
    *grid, n_luts = traveltimes.shape
    n_onsets, t_samples = lon.shape
    n_samples = t_samples - first_idx - last_idx
    if n_luts != n_onsets:
        raise ValueError(
            f"Mismatch between number of stations for data and LUT, "
            f"{n_onsets}:{n_luts}")
    if lon.size < n_samples + first_idx:
        raise ValueError("Data array smaller than coalescence array.")
    map4d = np.zeros(tuple(grid) + (n_samples,), dtype=np.float64)
    fns["stack"](lon, np.ascontiguousarray(traveltimes), map4d, first_idx,
                 last_idx, n_samples, n_onsets, int(available),
                 int(np.prod(grid)), int(threads))
    return map4d


def _scan_with(fns, map4d, threads):
    *grid, n_samples = map4d.shape
    n_nodes = int(np.prod(grid))
    max_coa = np.zeros(n_samples)
    max_norm = np.zeros(n_samples)
    idx = np.zeros(n_samples, dtype=np.int64)
    fns["scan"](np.ascontiguousarray(map4d), max_coa, max_norm, idx,
                n_samples, n_nodes, int(threads))
    return max_coa, max_norm, idx


def c_migrate(onsets, traveltimes, first_idx, last_idx, available, threads=1,
              prelogged=False):
    return _migrate_with(_port(), onsets, traveltimes, first_idx, last_idx,
                         available, threads, prelogged)


def c_find_max_coa(map4d, threads=1):
    return _scan_with(_port(), map4d, threads)


def ref_migrate(onsets, traveltimes, first_idx, last_idx, available,
                threads=1, prelogged=False):
    return _migrate_with(_ref(), onsets, traveltimes, first_idx, last_idx,
                         available, threads, prelogged)


def ref_find_max_coa(map4d, threads=1):
    return _scan_with(_ref(), map4d, threads)


def detect(onsets, traveltimes, first_idx, last_idx, available, threads=1,
           max_bytes=2 << 30, impl="port", prelogged=False):
    """
    migrate + find_max_coa without holding the whole 4-D volume: the scan is a
    per-sample reduction over nodes, so the time axis is cut into chunks whose
    volume fits ``max_bytes``; results are identical to the one-shot call
    (same per-sample node order for max, argmax and sum).
    """
    fns = _port() if impl == "port" else _ref()
    lon = np.ascontiguousarray(onsets if prelogged else log_onsets(onsets))
    tt = np.ascontiguousarray(traveltimes)
    *grid, n_rows = tt.shape
    n_nodes = int(np.prod(grid))
    t_samples = lon.shape[1]
    n_samples = t_samples - first_idx - last_idx
    chunk = int(max(1, min(n_samples, max_bytes // (8 * n_nodes))))
    max_coa = np.zeros(n_samples)
    max_norm = np.zeros(n_samples)
    idx = np.zeros(n_samples, dtype=np.int64)
    for k0 in range(0, n_samples, chunk):
        k1 = min(n_samples, k0 + chunk)
        vol = np.zeros((n_nodes, k1 - k0))
        fns["stack"](lon, tt, vol, first_idx + k0, t_samples - first_idx - k1,
                     k1 - k0, n_rows, int(available), n_nodes, int(threads))
        a, b, c = (np.zeros(k1 - k0), np.zeros(k1 - k0),
                   np.zeros(k1 - k0, dtype=np.int64))
        fns["scan"](vol, a, b, c, k1 - k0, n_nodes, int(threads))
        max_coa[k0:k1], max_norm[k0:k1], idx[k0:k1] = a, b, c
    return max_coa, max_norm, idx


# --------------------------------------------------------------------------
# NumPy restatement (portable spec; slow, small cases only)
# --------------------------------------------------------------------------
def np_migrate(onsets, traveltimes, first_idx, last_idx, available,
               prelogged=False):
    """migratelib.c:40-65 with lib.py:93-101 in front of it."""
    lon = onsets if prelogged else log_onsets(onsets)
    *grid, n_rows = traveltimes.shape
    n_nodes = int(np.prod(grid))
    t_samples = lon.shape[1]
    n_samples = t_samples - first_idx - last_idx
    tt = np.maximum(traveltimes.reshape(n_nodes, n_rows), 0).astype(np.int64)
    k = np.arange(n_samples, dtype=np.int64)[None, :]
    acc = np.zeros((n_nodes, n_samples))
    for r in range(n_rows):                       # ascending row order
        acc += lon[r][tt[:, r][:, None] + first_idx + k]
    return np.exp(acc / float(available)).reshape(tuple(grid) + (n_samples,))


def np_find_max_coa(map4d):
    """migratelib.c:85-111 (first maximum wins; sequential node-order sum)."""
    *grid, n_samples = map4d.shape
    vol = map4d.reshape(-1, n_samples)
    n_nodes = vol.shape[0]
    idx = np.argmax(vol, axis=0).astype(np.int64)      # first occurrence
    max_coa = vol[idx, np.arange(n_samples)]
    total = np.zeros(n_samples)
    for node in range(n_nodes):                        # sequential, as in C
        total += vol[node]
    return max_coa, max_coa * n_nodes / total, idx


def np_argmax_exp_rule(onsets, traveltimes, first_idx, last_idx, available, prelogged=False):
    """
    The reference's arg-max as its scalar-libm build computes it (migratelib.c:60-62 then :98-105):
    stacks in ascending row order, x = stack * (1 / available), a correctly rounded exp(x) and the
    FIRST node reaching the largest value.  exp: 80-bit arithmetic rounded to float64 for every
    node-sample, and -- round 6 -- for the nodes that can tie with a sample's largest x (within a few
    ulps of it) a 200-bit evaluation rounded once (mpmath): the 80-bit value rounded a second time is
    off by an ulp about once in a few thousand arguments, which the larger fixture near_ties_bricks
    caught on one sample of 1600 (x = 0.6018873029417819: ...994, not ...992).
    What the engine's opt-in ``tie_rule = 1`` is specified as (csrc/qm_ties.hpp); pinned on
    tests/golden/near_ties_scalar.npz and near_ties_bricks.npz (the reference's two C files built with
    -fno-tree-vectorize).
    """
    lon = onsets if prelogged else log_onsets(onsets)
    *grid, n_rows = traveltimes.shape
    n_nodes = int(np.prod(grid))
    n_samples = lon.shape[1] - first_idx - last_idx
    tt = np.maximum(traveltimes.reshape(n_nodes, n_rows), 0).astype(np.int64)
    k = np.arange(n_samples, dtype=np.int64)[None, :]
    acc = np.zeros((n_nodes, n_samples))
    for r in range(n_rows):                       # ascending row order
        acc += lon[r][tt[:, r][:, None] + first_idx + k]
    x = acc * (1.0 / float(available))            # the product the -Ofast builds form
    coa = np.exp(x.astype(np.longdouble)).astype(np.float64)
    out = np.argmax(coa, axis=0).astype(np.int64)
    try:
        import mpmath
    except ImportError:                            # (the 80-bit evaluation alone: see above)
        return out
    xmax = x.max(axis=0)
    near = x >= xmax - (1e-15 + 1e-15 * np.abs(xmax))
    with mpmath.workprec(200):
        for t in np.flatnonzero(near.sum(axis=0) > 1):
            nodes = np.flatnonzero(near[:, t])
            vals = [float(mpmath.exp(mpmath.mpf(float(x[n, t])))) for n in nodes]
            out[t] = nodes[int(np.argmax(vals))]   # (first maximum = lowest index)
    return out


# --------------------------------------------------------------------------
# Table serving (host-side NumPy in the reference)
# --------------------------------------------------------------------------
def np_decimate(grid, df):
    """Grid3D.decimate on one travel-time grid (quakemigrate/lut/lut.py:120-136)."""
    df = np.array(df, dtype=int)
    node_count = np.array(grid.shape)
    new_node_count = 1 + (node_count - 1) // df
    c1 = (node_count - df * (new_node_count - 1) - 1) // 2
    return grid[c1[0]::df[0], c1[1]::df[1], c1[2]::df[2]]


def np_serve_traveltimes(grids, sampling_rate):
    """LUT.serve_traveltimes (lut.py:536-538): stack on the last axis, rint, int32."""
    traveltimes = np.stack(list(grids), axis=-1)
    return np.rint(traveltimes * sampling_rate).astype(np.int32)


# --------------------------------------------------------------------------
# STA/LTA (lib.py:176-285 semantics: output pre-filled with ones / zeros)
# --------------------------------------------------------------------------
def _stalta(fn, signal, nsta, nlta, fill):
    head = np.empty(1, dtype=stalta_header_t)
    head[:] = (len(signal), nsta, nlta)
    signal = np.ascontiguousarray(signal, dtype=np.float64)
    onset = np.full(len(signal), fill, dtype=np.float64)
    fn(signal, head, onset)
    return onset


def c_overlapping_sta_lta(signal, nsta, nlta):
    return _stalta(_port()["overlapping"], signal, nsta, nlta, 1.0)


def c_centred_sta_lta(signal, nsta, nlta):
    return _stalta(_port()["centred"], signal, nsta, nlta, 1.0)


def c_recursive_sta_lta(signal, nsta, nlta):
    return _stalta(_port()["recursive"], signal, nsta, nlta, 0.0)


# --------------------------------------------------------------------------
# Onset stage (host-side NumPy around the C STA/LTA in the reference)
# --------------------------------------------------------------------------
def np_onset_stage(signals, trace_row, nsta, nlta, transform="energy",
                   position="classic", taper_pad=-1, min_onset_value=0.4,
                   stalta=None):
    """
    ``STALTAOnset._onset`` (quakemigrate/signal/onsets/stalta.py:515-546) with the taper
    windows of ``_trim_taper_pad`` (:579-581), then ``lib.migrate``'s clip + log (lib.py:93-94).
    ``stalta``: ``(overlapping_fn, centred_fn)`` taking (signal, nsta, nlta); defaults to the
    C port.  Returns ``(raw_onsets, log_onsets)``, shape (n_rows, T).
    """
    over, cent = stalta or (c_overlapping_sta_lta, c_centred_sta_lta)
    fn = over if position == "classic" else cent
    raw = []
    for row in range(len(nsta)):
        comps = []
        for tr in np.flatnonzero(np.asarray(trace_row) == row):
            x = signals[tr] ** 2 if transform == "energy" else np.abs(signals[tr])
            o = fn(x, int(nsta[row]), int(nlta[row]))
            if taper_pad >= 0:
                o[: (taper_pad + int(nlta[row]) - 1)] = 1.0
                o[-(int(nsta[row]) + taper_pad):] = 1.0
            comps.append(o)
        onset = np.sqrt(np.sum([c ** 2 for c in comps], axis=0) / len(comps))
        raw.append(np.clip(onset, min_onset_value, np.inf))
    raw = np.stack(raw, axis=0)
    return raw, np.log(np.clip(raw, 0.01, np.inf))


# --------------------------------------------------------------------------------------
# locate post-reductions: QuakeScan._calculate_location's array work
# --------------------------------------------------------------------------------------
def np_window_bounds(shape, centre, window):
    """``QuakeScan._mask3d`` bounds (scan.py:1066-1072): ``[lo, hi)`` per axis, clipped."""
    n = np.asarray(shape)
    c = np.asarray(centre)
    half = (window - 1) // 2
    return np.clip(c - half, 0, n), np.clip(c + half + 1, 0, n)


def np_gaufilt3d(map3d, sgm=0.8):
    """
    ``QuakeScan._gaufilt3d`` (scan.py:1008-1043) with ``util.gaussian_3d`` (util.py:76-116): a
    map-sized Gaussian, FFT-convolved in "same" mode, normalised; mirrored, convolved and
    normalised again.
    """
    from scipy.signal import fftconvolve

    map3d = np.asarray(map3d, dtype=np.float64)
    axes = [np.linspace(-(n - 1) / 2, (n - 1) / 2, n) for n in map3d.shape]
    gx, gy, gz = np.meshgrid(*axes, indexing="ij")
    flt = np.exp(-(gx * gx) / (2 * sgm * sgm) - (gy * gy) / (2 * sgm * sgm)
                 - (gz * gz) / (2 * sgm * sgm))
    out = map3d
    for _ in range(2):
        out = fftconvolve(out, flt, mode="same")
        out = out[::-1, ::-1, ::-1] / np.nanmax(out)
    return out


def np_covfit3d(coa_map, node_spacing, thresh=0.90):
    """
    ``QuakeScan._covfit3d`` (scan.py:939-1005) without the coordinate transform: returns
    ``(expectation xyz relative to the grid's lower-left corner, 3x3 covariance)``; the
    reference's uncertainty is ``sqrt(abs(diag(cov)))``.
    """
    coa_map = np.asarray(coa_map, dtype=np.float64)
    w = np.where(coa_map > thresh, coa_map, np.nan).ravel()
    total = np.nansum(w)
    idx = np.meshgrid(*[np.arange(n) for n in coa_map.shape], indexing="ij")
    pos = [g.ravel() * s for g, s in zip(idx, node_spacing)]
    mean = np.array([np.nansum(w * p) / total for p in pos])
    cov = np.zeros((3, 3))
    for a in range(3):
        for b in range(a, 3):
            if a == b:
                cov[a, a] = np.nansum(w * (pos[a] - mean[a]) ** 2) / total
            else:
                cov[a, b] = cov[b, a] = np.nansum(w * (pos[a] - mean[a]) * (pos[b] - mean[b])) / total
    return mean, cov


def np_gaufit3d(smoothed, thresh=0.0, win=7):
    """
    ``QuakeScan._gaufit3d`` (scan.py:844-936) in grid-index space: returns ``(location ijk
    (fractional), sigma in nodes (from the eigenvalues, scan.py:922-923), peak value)``; the
    reference's uncertainty is ``sigma * node_spacing``.
    """
    smoothed = np.asarray(smoothed, dtype=np.float64)
    peak = np.unravel_index(np.nanargmax(smoothed), smoothed.shape)
    lo, hi = np_window_bounds(smoothed.shape, peak, win)
    mask = np.zeros(smoothed.shape, dtype=bool)
    mask[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = True
    ix, iy, iz = np.where(mask & (smoothed > thresh))
    centred = smoothed - np.nanmean(smoothed)
    x, y, z = ix - peak[0], iy - peak[1], iz - peak[2]
    design = np.stack([x * x, y * y, z * z, x * y, x * z, y * z, x, y, z, np.ones(len(x))])
    rhs = -np.log(np.clip(centred[ix, iy, iz], 1e-300, np.inf))
    p = rhs @ np.linalg.pinv(design)
    g = -np.array([[2 * p[0], p[3], p[4]], [p[3], 2 * p[1], p[5]], [p[4], p[5], 2 * p[2]]])
    loc = np.linalg.inv(g) @ p[6:9]
    k = (p[9] - p[0] * loc[0] ** 2 - p[1] * loc[1] ** 2 - p[2] * loc[2] ** 2
         - p[3] * loc[0] * loc[1] - p[4] * loc[0] * loc[2] - p[5] * loc[1] * loc[2])
    m = np.array([[p[0], p[3] / 2, p[4] / 2], [p[3] / 2, p[1], p[5] / 2],
                  [p[4] / 2, p[5] / 2, p[2]]])
    egv, _ = np.linalg.eig(m)
    sigma = np.sqrt(0.5 / np.clip(np.abs(egv), 1e-10, np.inf)) / 2
    return loc + np.array(peak), sigma, np.exp(-k)


def np_splineloc(coa_map, win=5, upscale=10):
    """
    ``QuakeScan._splineloc`` (scan.py:736-841) in grid-index space: cubic RBF through the
    ``win``^3 nodes around the maximum, evaluated ``upscale`` times finer; the gridded maximum
    if the window crosses the edge of the grid or the interpolated peak leaves the window.
    """
    from scipy.interpolate import Rbf

    coa_map = np.asarray(coa_map, dtype=np.float64)
    peak = np.array(np.unravel_index(np.nanargmax(coa_map), coa_map.shape))
    lo, hi = np_window_bounds(coa_map.shape, peak, win)
    ext = hi - lo
    if not (ext[0] == ext[1] == ext[2]):
        return peak.astype(np.float64)
    sub = coa_map[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
    # the reference's meshgrid is "xy"-indexed, so its axis-0 coordinate runs along axis 1 of
    # the window (scan.py:780-787): reproduce that pairing, not the natural one
    coarse = [np.linspace(0, n - 1, n) for n in sub.shape]
    cx, cy, cz = np.meshgrid(*coarse)
    rbf = Rbf(cx.ravel(), cy.ravel(), cz.ravel(), sub.ravel(), function="cubic")
    fine = [np.linspace(0, n - 1, (n - 1) * upscale + 1) for n in sub.shape]
    fx, fy, fz = np.meshgrid(*fine)
    dense = rbf(fx.ravel(), fy.ravel(), fz.ravel()).reshape(fx.shape)
    best = np.array(np.unravel_index(np.nanargmax(dense), dense.shape)) / upscale + lo
    if np.any(np.abs(peak - best) > (win - 1) // 2):
        return peak.astype(np.float64)
    return best
