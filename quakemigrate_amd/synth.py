# -*- coding: utf-8 -*-
"""
Deterministic synthetic inputs for the migrate / find_max_coa path.

Recipe = BASELINE.md "Configs / synthetic inputs" (SURVEY.md section 8d):
homogeneous-velocity travel-time tables ``tt = dist / v`` (the same arithmetic as
the reference's ``_compute_homogeneous``, quakemigrate/lut/create_lut.py:256-265),
stations uniform over the x-y footprint at z = 0, first half of the rows P
(vp = 5.0 km/s), second half S (vs = 2.9 km/s), ``rint(tt * sampling_rate)`` to
int32 exactly as ``LUT.serve_traveltimes`` does (quakemigrate/lut/lut.py:538).
Raw onsets are ``clip(lognormal(0, 0.5), 0.4, inf)`` with a few injected events
whose arrivals follow the table, so the argmax is not trivial.

NumPy only (runs on any host; identical integers everywhere because sqrt and
division are correctly rounded).
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# name -> (nx, ny, nz, spacing_km, n_rows, n_samples, sampling_rate)
CONFIGS = {
    # Icequake_Iceland-sized geometry (examples/Icequake_Iceland/iceland_lut.py:37-50,
    # iceland_detect.py:45,62): 25 m cells, 250 Hz, 2.5 s timestep, 12 stations x P,S
    "C1": dict(grid=(71, 64, 57), spacing=0.025, rows=24, n_samples=625,
               rate=250.0, vp=3.630, vs=1.833, fsmp=413, paired=True),
    "C2": dict(grid=(101, 101, 51), spacing=1.0, rows=20, n_samples=6000,
               rate=50.0, vp=5.0, vs=2.9, fsmp=580, paired=False),
    "C3": dict(grid=(201, 201, 101), spacing=0.5, rows=30, n_samples=6000,
               rate=50.0, vp=5.0, vs=2.9, fsmp=580, paired=False),
    "C4": dict(grid=(401, 401, 201), spacing=0.5, rows=60, n_samples=12000,
               rate=50.0, vp=5.0, vs=2.9, fsmp=580, paired=False),
    # The sizes the reference's own examples run detect() at ("Context"): both
    # decimate their LUT by [2, 2, 2] first, 50 Hz, P and S of every station.
    # E1 Volcanotectonic_Iceland: 0.5 km nodes -> 1 km, 12 stations, timestep 300 s
    # (examples/Volcanotectonic_Iceland/dike_intrusion_lut.py:42-44, dike_intrusion_detect.py:43,65)
    "E1": dict(grid=(29, 29, 19), spacing=1.0, rows=24, n_samples=15000,
               rate=50.0, vp=5.0, vs=2.9, fsmp=580, paired=True),
    # E2 Askja_Iceland_VT-DLP: 1 km nodes -> 2 km, 23 stations, timestep 60 s
    # (examples/Askja_Iceland_VT-DLP/askja_lut.py:42-44, askja_detect.py:44,66)
    "E2": dict(grid=(36, 31, 21), spacing=2.0, rows=46, n_samples=3000,
               rate=50.0, vp=5.0, vs=2.9, fsmp=580, paired=True),
    # ... and the same LUT NOT decimated (1 km nodes, askja_lut.py:42-44): what one would feed a GPU
    "E2F": dict(grid=(72, 62, 41), spacing=1.0, rows=46, n_samples=3000,
                rate=50.0, vp=5.0, vs=2.9, fsmp=580, paired=True),
    # locate-style window on the C3 grid: 4 * marginal_window(2 s) * 50 Hz + 1
    "C3L": dict(grid=(201, 201, 101), spacing=0.5, rows=30, n_samples=401,
                rate=50.0, vp=5.0, vs=2.9, fsmp=580, paired=False),
}
CONFIG_IDS = {"C1": 1, "C2": 2, "C3": 3, "C4": 4, "C3L": 3, "E1": 11, "E2": 12, "E2F": 12}
BASE_SEED = 20260927


def station_positions(rng, grid, spacing, n_stations):
    nx, ny, _ = grid
    xy = rng.uniform([0.0, 0.0], [(nx - 1) * spacing, (ny - 1) * spacing],
                     size=(n_stations, 2))
    return np.concatenate([xy, np.zeros((n_stations, 1))], axis=1)


def homogeneous_lut(grid, spacing, stations_xyz, velocities, rate,
                    x_range=None):
    """
    int32 table, shape (nx, ny, nz, n_rows); row r = station r at velocity r.

    ``x_range=(x0, x1)`` builds only the x-planes [x0, x1) (a contiguous range of
    flat node indices) -- used by the sharded path so that no rank ever builds
    the whole table.
    """
    nx, ny, nz = grid
    x0, x1 = (0, nx) if x_range is None else x_range
    gx = (np.arange(x0, x1, dtype=np.float64) * spacing)[:, None, None]
    gy = (np.arange(ny, dtype=np.float64) * spacing)[None, :, None]
    gz = (np.arange(nz, dtype=np.float64) * spacing)[None, None, :]
    out = np.empty((x1 - x0, ny, nz, len(velocities)), dtype=np.int32)
    for r, (xyz, v) in enumerate(zip(stations_xyz, velocities)):
        dist = np.sqrt((gx - xyz[0]) ** 2 + (gy - xyz[1]) ** 2
                       + (gz - xyz[2]) ** 2)
        out[..., r] = np.rint(dist / v * rate).astype(np.int32)
    return out


def synthetic_onsets(rng, n_rows, t_samples, event_arrivals=(), amplitude=8.0,
                     sigma=5.0):
    """
    Raw (un-logged) onset rows, shape (n_rows, t_samples), values >= 0.4.
    ``event_arrivals``: iterable of int arrays (n_rows,) -- sample index in the
    row at which each row sees the event; a Gaussian bump is added there.
    """
    on = np.clip(rng.lognormal(mean=0.0, sigma=0.5, size=(n_rows, t_samples)),
                 0.4, np.inf)
    k = np.arange(t_samples, dtype=np.float64)
    for arr in event_arrivals:
        for r in range(n_rows):
            on[r] += amplitude * np.exp(-0.5 * ((k - float(arr[r])) / sigma) ** 2)
    return on


@dataclass
class Case:
    name: str
    grid: tuple
    traveltimes: np.ndarray     # int32 (nx_local, ny, nz, S)
    onsets: np.ndarray          # raw float64 (S, T)
    fsmp: int
    lsmp: int
    n_samples: int
    available: int
    x_range: tuple              # planes of the full grid held in `traveltimes`
    event_nodes: list
    stations: np.ndarray
    velocities: np.ndarray

    @property
    def n_nodes_total(self):
        return int(np.prod(self.grid))


def make_case(name, step=0, x_range=None, n_samples=None, grid=None, rows=None,
              n_events=3, quiet=False, table=True) -> Case:
    """
    Build configuration ``name`` (C1..C4, C3L) for timestep ``step``.  The table
    depends only on the configuration; the onsets also on ``step``.
    ``grid`` / ``rows`` / ``n_samples`` override the named sizes (tests use
    shrunken variants of the same recipe).  ``table=False`` skips building the
    travel-time table (``traveltimes`` is None): further timesteps of a configuration
    whose table the caller already holds.
    """
    cfg = dict(CONFIGS[name])
    if grid is not None:
        cfg["grid"] = tuple(grid)
    if rows is not None:
        cfg["rows"] = int(rows)
    if n_samples is not None:
        cfg["n_samples"] = int(n_samples)
    g, S, ns = cfg["grid"], cfg["rows"], cfg["n_samples"]
    rng = np.random.default_rng(BASE_SEED + CONFIG_IDS[name])
    if cfg["paired"]:           # S/2 stations, each seen as P then S
        st = station_positions(rng, g, cfg["spacing"], S // 2)
        st = np.concatenate([st, st], axis=0)
    else:
        st = station_positions(rng, g, cfg["spacing"], S)
    vel = np.array([cfg["vp"]] * (S // 2) + [cfg["vs"]] * (S - S // 2))
    # largest delay: farthest grid corner from each station
    corners = np.array([[x, y, z] for x in (0, g[0] - 1) for y in (0, g[1] - 1)
                        for z in (0, g[2] - 1)], dtype=np.float64) * cfg["spacing"]
    far = np.sqrt(((corners[None, :, :] - st[:, None, :]) ** 2).sum(-1)).max(1)
    tt_max = int(np.rint(far / vel * cfg["rate"]).max())
    fsmp, lsmp = int(cfg["fsmp"]), tt_max + 100
    t_samples = fsmp + ns + lsmp
    tt = homogeneous_lut(g, cfg["spacing"], st, vel, cfg["rate"], x_range) if table else None

    rng_on = np.random.default_rng([BASE_SEED + CONFIG_IDS[name], 1 + step])
    nodes, arrivals = [], []
    if not quiet:
        for _ in range(n_events):
            ijk = [int(rng_on.integers(0, d)) for d in g]
            t0 = int(rng_on.integers(ns // 10, max(ns // 10 + 1, ns - ns // 10)))
            xyz = np.array(ijk, dtype=np.float64) * cfg["spacing"]
            d = np.sqrt(((st - xyz[None, :]) ** 2).sum(-1))
            tt_node = np.rint(d / vel * cfg["rate"]).astype(np.int64)
            nodes.append((tuple(ijk), t0))
            arrivals.append(fsmp + t0 + tt_node)
    if quiet:
        on = np.full((S, t_samples), 0.4)
    else:
        on = synthetic_onsets(rng_on, S, t_samples, arrivals)
    xr = (0, g[0]) if x_range is None else tuple(x_range)
    return Case(name, g, tt, on, fsmp, lsmp, ns, S, xr, nodes, st, vel)
