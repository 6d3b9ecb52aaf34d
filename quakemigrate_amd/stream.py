# -*- coding: utf-8 -*-
"""
Continuous detect sweep with copies overlapped with compute (BASELINE.json configs[4]:
"continuous 24 h synthetic stream ... overlapped H2D copy + compute on HIP streams").

The reference's ``QuakeScan._continuous_compute`` (quakemigrate/signal/scan.py:407-470; the loop
:434-448) walks the timesteps serially: read -> onsets -> migrate -> find_max_coa -> append.  The
hot-path part of that loop is the library's native pipeline (``qm_stream_*``, include/qmhip.h part 3,
csrc/qm_stream.hip): a pinned ring of ``depth`` slots of ``steps_per_launch`` timesteps, H2D on a copy
stream, one fused-detect launch per slot, one packed D2H per launch, HIP events for the ordering.
This class is the thin caller: it pushes windows, pops results in order, and nothing else -- no
torch, no per-step staging in Python (round 4's Python loop cost the example-sized grids 30-40 % of
their kernel rate once the copies were inside the clock).
"""

from __future__ import annotations

import ctypes

import numpy as np

from quakemigrate_amd.core import lib as _lib

_qm = _lib.qmlib


class StreamingDetector:
    """
    Parameters
    ----------
    engine : quakemigrate_amd.core.Engine with the travel-time table resident.
    n_rows, t_samples : shape of every onset window (rows x samples, float64, already
        ``log(clip(., 0.01))``).
    fsmp, lsmp, available : as in ``Engine.detect``.
    n_nodes_total : node count of the full grid (normalisation).
    depth : slots of the ring = launches that may be in flight or un-popped (>= 2).
    steps_per_launch : timesteps stacked by ONE launch (``qm_engine_detect_batch``).  Timesteps are
        independent given their onsets, so K of them can share a launch: on the grids the
        reference's examples use (1e4 - 3e5 nodes) one timestep is a few workgroup rounds and a
        fraction of a millisecond, and K steps per launch are what fills the GPU and amortises the
        launch, the combine and the copies' latencies.  Results are identical step for step.
    device : accepted for compatibility with round 4's signature; the engine's device is used.
    """

    def __init__(self, engine, n_rows, t_samples, fsmp, lsmp, available, n_nodes_total=None,
                 depth=2, device=None, steps_per_launch=1):
        if engine.n_rows is None:
            raise _lib.QMHipError("no travel-time table resident: call load_lut first")
        if int(n_rows) != engine.n_rows:
            raise ValueError("Mismatch between number of stations for data and LUT, "
                             f"{int(n_rows)}:{engine.n_rows}")
        self.engine = engine
        self.n_rows, self.t_samples = int(n_rows), int(t_samples)
        self.fsmp, self.lsmp, self.available = int(fsmp), int(lsmp), int(available)
        self.n_samples = self.t_samples - self.fsmp - self.lsmp
        self.depth = max(2, int(depth))
        self.k = max(1, int(steps_per_launch))
        total = engine.n_nodes if n_nodes_total is None else int(n_nodes_total)
        h = ctypes.c_void_p()
        _lib._check(_qm.qm_stream_create(engine._h, self.t_samples, self.fsmp, self.lsmp,
                                         self.available, total, self.k, self.depth, ctypes.byref(h)))
        self._h = h
        import weakref

        self._finalizer = weakref.finalize(self, _qm.qm_stream_destroy, h)

    def close(self):
        self._finalizer()

    # -- the three calls --------------------------------------------------------------
    def push(self, window):
        """One timestep's log-onsets (n_rows, t_samples) into the pipeline.  False: every slot holds
        results that have not been popped (``pop`` first, then push again)."""
        w = np.ascontiguousarray(window, dtype=np.float64)
        if w.shape != (self.n_rows, self.t_samples):
            raise ValueError(f"window of shape {w.shape}, the stream takes {(self.n_rows, self.t_samples)}")
        rc = _qm.qm_stream_push(self._h, w.ctypes.data_as(ctypes.c_void_p))
        if rc == 2:
            return False
        _lib._check(rc)
        return True

    def flush(self):
        _lib._check(_qm.qm_stream_flush(self._h))

    def pending(self):
        """(timesteps launched and not yet popped, timesteps pushed into a launch that has not gone out)"""
        a, b = ctypes.c_int32(), ctypes.c_int32()
        _lib._check(_qm.qm_stream_pending(self._h, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def pop(self, n_steps=1):
        """The next ``n_steps`` timesteps' ``(max_coa, max_norm_coa, max_coa_idx)``, each
        ``(n_steps, n_samples)``; blocks until their launch has finished."""
        n, ns = int(n_steps), self.n_samples
        a, b = np.empty((n, ns)), np.empty((n, ns))
        c = np.empty((n, ns), dtype=np.int64)
        _lib._check(_qm.qm_stream_pop(self._h, n, a.ctypes.data_as(ctypes.c_void_p),
                                      b.ctypes.data_as(ctypes.c_void_p), c.ctypes.data_as(ctypes.c_void_p)))
        return a, b, c

    # -- the loop -----------------------------------------------------------------------
    def run(self, windows, on_result=None):
        """
        ``windows``: iterable of float64 arrays (n_rows, t_samples), one per timestep, already
        logged.  Returns a list of ``(max_coa, max_norm_coa, max_coa_idx)`` NumPy triples (or
        calls ``on_result(step, triple)`` and returns the number of steps).
        """
        results = []
        step = 0

        def take(n):
            nonlocal step
            a, b, c = self.pop(n)
            for j in range(n):
                triple = (a[j], b[j], c[j])
                if on_result is None:
                    results.append(triple)
                else:
                    on_result(step, triple)
                step += 1

        for w in windows:
            while not self.push(w):
                take(min(self.k, self.pending()[0]))     # the oldest launch's timesteps
        self.flush()
        left = self.pending()[0]
        while left > 0:
            n = min(self.k, left)
            take(n)
            left -= n
        return results if on_result is None else step
