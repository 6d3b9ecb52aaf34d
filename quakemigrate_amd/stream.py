# -*- coding: utf-8 -*-
"""
Continuous detect sweep with copies overlapped with compute (BASELINE.json configs[4]:
"continuous 24 h synthetic stream ... overlapped H2D copy + compute on HIP streams").

The reference's ``QuakeScan._continuous_compute`` (quakemigrate/signal/scan.py:407-470)
walks the timesteps serially: read -> onsets -> migrate -> find_max_coa -> append.  Here the
hot-path part of that loop is pipelined on two HIP streams per GPU:

    copy stream    : H2D of the log-onsets of step i+1   (pinned, double-buffered)
    compute stream : fused detect of step i, D2H of its three series (pinned)

Timesteps are independent given their onsets ("long-context"), so the only
ordering is buffer reuse, expressed with HIP events.  PyTorch supplies the streams, events and
pinned memory (plumbing); the compute is the engine's HIP kernels.
"""

from __future__ import annotations

import numpy as np
import torch


class StreamingDetector:
    """
    Parameters
    ----------
    engine : quakemigrate_amd.core.Engine with the travel-time table resident.
    n_rows, t_samples : shape of every onset window (rows x samples, float64, already
        ``log(clip(., 0.01))``).
    fsmp, lsmp, available : as in ``Engine.detect``.
    n_nodes_total : node count of the full grid (normalisation).
    depth : number of onset windows in flight (>= 2).
    """

    def __init__(self, engine, n_rows, t_samples, fsmp, lsmp, available, n_nodes_total=None,
                 depth=2, device=None):
        self.engine = engine
        self.fsmp, self.lsmp, self.available = int(fsmp), int(lsmp), int(available)
        self.n_samples = int(t_samples) - self.fsmp - self.lsmp
        self.n_nodes_total = n_nodes_total
        self.device = torch.device("cuda", engine.device) if device is None else device
        self.depth = max(2, int(depth))
        self.copy_stream = torch.cuda.Stream(self.device)
        self.compute_stream = torch.cuda.Stream(self.device)
        shape = (int(n_rows), int(t_samples))
        ns = self.n_samples
        self.h_on = [torch.empty(shape, dtype=torch.float64).pin_memory()
                     for _ in range(self.depth)]
        # NumPy views of the pinned input buffers: the host-side copy into them is a plain
        # single-threaded memcpy.  (torch's CPU copy_ fans a 1-2 MB copy out over its OpenMP
        # pool; after a blocking wait the pool has to be woken up, which on a 256-core host costs
        # tens of milliseconds every few steps -- 4x the whole step at C2 size.)
        self.h_on_np = [t.numpy() for t in self.h_on]
        self.d_on = [torch.empty(shape, dtype=torch.float64, device=self.device)
                     for _ in range(self.depth)]
        self.d_out = [(torch.empty(ns, dtype=torch.float64, device=self.device),
                       torch.empty(ns, dtype=torch.float64, device=self.device),
                       torch.empty(ns, dtype=torch.int64, device=self.device))
                      for _ in range(self.depth)]
        self.h_out = [(torch.empty(ns, dtype=torch.float64).pin_memory(),
                       torch.empty(ns, dtype=torch.float64).pin_memory(),
                       torch.empty(ns, dtype=torch.int64).pin_memory())
                      for _ in range(self.depth)]
        self.copied = [torch.cuda.Event() for _ in range(self.depth)]      # H2D landed
        self.consumed = [torch.cuda.Event() for _ in range(self.depth)]    # kernel read it
        self.done = [torch.cuda.Event() for _ in range(self.depth)]        # outputs on host

    def run(self, windows, on_result=None):
        """
        ``windows``: iterable of float64 arrays (n_rows, t_samples), one per timestep, already
        logged.  Returns a list of ``(max_coa, max_norm_coa, max_coa_idx)`` NumPy triples (or
        calls ``on_result(step, triple)`` and returns the number of steps).
        """
        eng = self.engine
        eng.set_stream(self.compute_stream.cuda_stream)
        results = []
        pending = []                                   # (step, slot) whose outputs are in flight
        it = iter(windows)

        def stage(step, slot, array):
            # the slot's previous contents must have been consumed by its kernel
            if step >= self.depth:
                self.consumed[slot].synchronize()
            np.copyto(self.h_on_np[slot], array)
            with torch.cuda.stream(self.copy_stream):
                self.d_on[slot].copy_(self.h_on[slot], non_blocking=True)
                self.copied[slot].record(self.copy_stream)

        def collect(step, slot):
            self.done[slot].synchronize()
            triple = tuple(t.numpy().copy() for t in self.h_out[slot])
            if on_result is None:
                results.append(triple)
            else:
                on_result(step, triple)

        nxt = next(it, None)
        step = 0
        if nxt is not None:
            stage(0, 0, nxt)
        while nxt is not None:
            slot = step % self.depth
            cur, nxt = nxt, next(it, None)
            if nxt is not None:                        # copy of step+1 overlaps compute of step
                stage(step + 1, (step + 1) % self.depth, nxt)
            if len(pending) >= self.depth:             # the slot's host outputs must be free
                collect(*pending.pop(0))
            with torch.cuda.stream(self.compute_stream):
                self.compute_stream.wait_event(self.copied[slot])
                eng.detect(self.d_on[slot], self.fsmp, self.lsmp, self.available,
                           n_nodes_total=self.n_nodes_total, out=self.d_out[slot])
                self.consumed[slot].record(self.compute_stream)
                for h, d in zip(self.h_out[slot], self.d_out[slot]):
                    h.copy_(d, non_blocking=True)
                self.done[slot].record(self.compute_stream)
            pending.append((step, slot))
            step += 1
        while pending:
            collect(*pending.pop(0))
        eng.set_stream(None)
        return results if on_result is None else step
