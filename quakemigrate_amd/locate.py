# -*- coding: utf-8 -*-
"""
Event location from the marginalised coalescence map -- the counterpart of
``QuakeScan._calculate_location`` (quakemigrate/signal/scan.py:696-733).

The sweeps over the whole 3-D map run on the GPU (``Engine.locate_fits``): normalisation by
the maximum, the two-pass Gaussian smoothing of ``_gaufilt3d`` (scan.py:1008-1043), the
thresholded moments of ``_covfit3d`` (scan.py:939-1005).  What is left on the host is the
algebra on the two small windows the engine hands back: the 10-parameter log-quadratic least
squares of ``_gaufit3d`` on (at most) 7x7x7 smoothed values (scan.py:844-936) and, of the cubic
radial-basis interpolation of ``_splineloc`` on 5x5x5 values (scan.py:736-841), the 125 x 125
solve for its weights; the interpolant's 41^3 values and their maximum go back to the GPU
(``Engine.rbf_peak``).

Everything is returned in grid-index / grid-xyz space.  Pass ``lut`` (anything with the
reference's ``index2coord`` / ``coord2grid`` / ``ll_corner``, lut/lut.py:174-243) to get the
coordinates the reference stores on the event.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class LocationFits:
    map_max: float                  # nanmax of the marginalised map (the normalisation)
    peak: np.ndarray                # ijk of the maximum of the normalised map
    spline: np.ndarray              # fractional ijk, _splineloc
    gaussian: np.ndarray            # fractional ijk, _gaufit3d
    gaussian_sigma: np.ndarray      # nodes; uncertainty = sigma * node_spacing (scan.py:934)
    gaussian_peak_value: float
    expectation: np.ndarray         # xyz relative to ll_corner, _covfit3d
    covariance: np.ndarray          # 3x3
    node_spacing: np.ndarray

    @property
    def gaussian_uncertainty(self):
        return self.gaussian_sigma * self.node_spacing

    @property
    def covariance_uncertainty(self):
        return np.diag(np.sqrt(np.abs(self.covariance)))        # scan.py:1003

    def coordinates(self, lut):
        """Spline, Gaussian and covariance locations through the reference LUT's transforms."""
        spline = lut.index2coord([list(self.spline)])[0]
        gaussian = lut.index2coord([list(self.gaussian)])[0]
        covariance = lut.coord2grid(lut.ll_corner + self.expectation, inverse=True)[0]
        return spline, gaussian, covariance


def _bounds(shape, centre, window):
    n = np.asarray(shape)
    c = np.asarray(centre)
    half = (window - 1) // 2
    return np.clip(c - half, 0, n), np.clip(c + half + 1, 0, n)


def gaussian_from_window(window, smoothed_mean, peak, shape, thresh=0.0):
    """
    ``_gaufit3d`` on the 7x7x7 ``window`` of the smoothed map centred on ``peak`` (NaN outside
    the grid).  Returns ``(location ijk, sigma in nodes, fitted peak value)``.
    """
    win = window.shape[0]
    half = (win - 1) // 2
    lo, hi = _bounds(shape, peak, win)
    a0, a1 = lo - (np.asarray(peak) - half), hi - (np.asarray(peak) - half)
    sub = window[a0[0]:a1[0], a0[1]:a1[1], a0[2]:a1[2]]
    ia, ib, ic = np.where(sub > thresh)
    x, y, z = ia + a0[0] - half, ib + a0[1] - half, ic + a0[2] - half
    design = np.stack([x * x, y * y, z * z, x * y, x * z, y * z, x, y, z, np.ones(len(x))])
    rhs = -np.log(np.clip(sub[ia, ib, ic] - smoothed_mean, 1e-300, np.inf))
    p = rhs @ np.linalg.pinv(design)
    g = -np.array([[2 * p[0], p[3], p[4]], [p[3], 2 * p[1], p[5]], [p[4], p[5], 2 * p[2]]])
    loc = np.linalg.inv(g) @ p[6:9]
    k = (p[9] - p[0] * loc[0] ** 2 - p[1] * loc[1] ** 2 - p[2] * loc[2] ** 2
         - p[3] * loc[0] * loc[1] - p[4] * loc[0] * loc[2] - p[5] * loc[1] * loc[2])
    m = np.array([[p[0], p[3] / 2, p[4] / 2], [p[3] / 2, p[1], p[5] / 2],
                  [p[4] / 2, p[5] / 2, p[2]]])
    egv, _ = np.linalg.eig(m)
    sigma = np.sqrt(0.5 / np.clip(np.abs(egv), 1e-10, np.inf)) / 2
    return loc + np.asarray(peak), sigma, float(np.exp(-k))


def _cubic_rbf_weights(sub):
    """
    Weights of ``scipy.interpolate.Rbf(x, y, z, d, function="cubic")`` (smooth = 0, Euclidean
    norm) through the cube ``sub``: ``solve(|ci - cj|^3, d)``, with the reference's (default,
    "xy") meshgrid pairing -- the value ``sub[a, b, c]`` sits at ``(x, y, z) = (b, a, c)``
    (scan.py:777-797).  Returned in the shape of ``sub``.
    """
    n = sub.shape[0]
    c = np.arange(n, dtype=np.float64)
    a, b, cc = np.meshgrid(c, c, c, indexing="ij")                     # value index (a, b, c)
    centres = np.stack([b.ravel(), a.ravel(), cc.ravel()], axis=1)     # its (x, y, z)
    diff = centres[:, None, :] - centres[None, :, :]
    r = np.sqrt(((diff[..., 0] ** 2 + diff[..., 1] ** 2) + diff[..., 2] ** 2))
    return np.linalg.solve(r ** 3, sub.ravel()).reshape(n, n, n)


def _cubic_rbf_on_grid(sub, upscale):
    """
    ``scipy.interpolate.Rbf(x, y, z, d, function="cubic")`` (smooth = 0, Euclidean norm) through
    the cube ``sub`` and its evaluation on the ``upscale``-times finer grid, spelled out: weights
    from ``solve(|ci - cj|^3, d)``, interpolant ``sum_i w_i |p - ci|^3``.  Same algebra and the
    same (default, "xy") meshgrid pairing as the reference's call (scan.py:777-804: the value
    ``sub[a, b, c]`` sits at ``(x, y, z) = (b, a, c)`` and ``dense[i, j, k]`` is evaluated at
    ``(j, i, k) / upscale``), but the point-to-centre distances are built per axis -- both sets
    are tensor grids -- which makes it ~50x faster than the SciPy class on the 41^3 points.
    """
    n = sub.shape[0]
    c = np.arange(n, dtype=np.float64)
    weights = _cubic_rbf_weights(sub)
    f = np.linspace(0, n - 1, (n - 1) * upscale + 1)
    d2 = (f[:, None] - c[None, :]) ** 2                                # [fine, coarse]
    # r2[i, j, k, a, b, c] = (x_j - b)^2 + (y_i - a)^2 + (z_k - c)^2, one i-slab at a time
    m = len(f)
    w = weights.ravel()
    out = np.empty((m, m, m))
    for i in range(m):
        r2 = ((d2[None, :, None, None, :, None] + d2[i][None, None, None, :, None, None])
              + d2[None, None, :, None, None, :])[0]                    # [j, k, a, b, c]
        r2 = r2.reshape(m * m, n * n * n)
        out[i] = ((r2 * np.sqrt(r2)) @ w).reshape(m, m)
    return out


def spline_from_window(window, peak, shape, upscale=10, engine=None):
    """
    ``_splineloc`` on the 5x5x5 ``window`` of the normalised map centred on ``peak``: the
    sub-node maximum of a cubic RBF through the window, or the gridded maximum when the clipped
    window is not a cube or the interpolated maximum leaves it (scan.py:772-839).  With an
    ``engine`` the interpolant's 41^3 values and their maximum are evaluated on the GPU
    (``Engine.rbf_peak``; the 125 x 125 solve for the weights stays here), otherwise in NumPy.
    The GPU sums the 125 centres sequentially, SciPy uses a BLAS dot: the 41^3 values agree to
    ~1e-14 relative, so the two pick a different fine-grid point only if the two largest values are
    closer than that -- both are then equally valid maxima of the interpolant
    (tests: test_spline_location_on_device_equals_scipy_rbf_on_map_windows).
    """
    win = window.shape[0]
    half = (win - 1) // 2
    peak = np.asarray(peak)
    lo, hi = _bounds(shape, peak, win)
    ext = hi - lo
    if not (ext[0] == ext[1] == ext[2]):
        return peak.astype(np.float64)
    a0, a1 = lo - (peak - half), hi - (peak - half)
    sub = window[a0[0]:a1[0], a0[1]:a1[1], a0[2]:a1[2]]
    if engine is not None:
        _, fine = engine.rbf_peak(_cubic_rbf_weights(sub), upscale)
        best = np.array(fine) / upscale + lo
    else:
        dense = _cubic_rbf_on_grid(sub, upscale)
        best = np.array(np.unravel_index(np.nanargmax(dense), dense.shape)) / upscale + lo
    if np.any(np.abs(peak - best) > half):
        return peak.astype(np.float64)
    return best


def calculate_location(engine, coa_map, node_spacing, sgm=0.8, cov_thresh=0.90,
                       norm_out=None, smoothed_out=None):
    """
    All three locations of ``_calculate_location`` for one marginalised map (host array or
    device tensor, e.g. the output of ``Engine.marginal_map``).  ``norm_out`` receives
    ``coa_map / nanmax(coa_map)``, what the reference returns and plots.
    """
    node_spacing = np.asarray(node_spacing, dtype=np.float64)
    shape = tuple(int(v) for v in coa_map.shape)
    dev = engine.locate_fits(coa_map, node_spacing, sgm=sgm, cov_thresh=cov_thresh,
                             norm_out=norm_out, smoothed_out=smoothed_out)
    gaussian, sigma, value = gaussian_from_window(dev["gaussian_window"], dev["smoothed_mean"],
                                                  dev["smoothed_peak"], shape)
    spline = spline_from_window(dev["spline_window"], dev["peak"], shape, engine=engine)
    return LocationFits(map_max=dev["map_max"], peak=dev["peak"], spline=spline,
                        gaussian=gaussian, gaussian_sigma=sigma, gaussian_peak_value=value,
                        expectation=dev["expectation"], covariance=dev["covariance"],
                        node_spacing=node_spacing)
