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
    depth : number of launches in flight (>= 2).
    steps_per_launch : timesteps stacked by ONE launch (``Engine.detect_batch``).  Timesteps are
        independent given their onsets, so K of them can share a launch: on the grids the
        reference's examples use (1e4 - 3e5 nodes) one timestep is a few workgroup rounds and a
        fraction of a millisecond, and K steps per launch are what fills the GPU and amortises the
        launch, the combine and the copies' latencies.  Results are identical step for step.
    """

    def __init__(self, engine, n_rows, t_samples, fsmp, lsmp, available, n_nodes_total=None,
                 depth=2, device=None, steps_per_launch=1):
        self.engine = engine
        self.fsmp, self.lsmp, self.available = int(fsmp), int(lsmp), int(available)
        self.n_samples = int(t_samples) - self.fsmp - self.lsmp
        self.n_nodes_total = n_nodes_total
        self.device = torch.device("cuda", engine.device) if device is None else device
        self.depth = max(2, int(depth))
        self.k = max(1, int(steps_per_launch))
        self.copy_stream = torch.cuda.Stream(self.device)
        self.compute_stream = torch.cuda.Stream(self.device)
        shape = (self.k, int(n_rows), int(t_samples))
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
        self.d_out = [(torch.empty((self.k, ns), dtype=torch.float64, device=self.device),
                       torch.empty((self.k, ns), dtype=torch.float64, device=self.device),
                       torch.empty((self.k, ns), dtype=torch.int64, device=self.device))
                      for _ in range(self.depth)]
        self.h_out = [(torch.empty((self.k, ns), dtype=torch.float64).pin_memory(),
                       torch.empty((self.k, ns), dtype=torch.float64).pin_memory(),
                       torch.empty((self.k, ns), dtype=torch.int64).pin_memory())
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
        pending = []                                   # (first step, slot, steps) in flight
        it = iter(windows)

        def take():
            """up to k windows from the iterator"""
            batch = []
            for w in it:
                batch.append(w)
                if len(batch) == self.k:
                    break
            return batch

        def stage(launch, slot, batch):
            # the slot's previous contents must have been consumed by its kernel
            if launch >= self.depth:
                self.consumed[slot].synchronize()
            for j, array in enumerate(batch):
                np.copyto(self.h_on_np[slot][j], array)
            with torch.cuda.stream(self.copy_stream):
                n = len(batch)
                self.d_on[slot][:n].copy_(self.h_on[slot][:n], non_blocking=True)
                self.copied[slot].record(self.copy_stream)

        def collect(first, slot, n):
            self.done[slot].synchronize()
            for j in range(n):
                triple = tuple(t.numpy()[j].copy() for t in self.h_out[slot])
                if on_result is None:
                    results.append(triple)
                else:
                    on_result(first + j, triple)

        nxt = take()
        launch, step = 0, 0
        if nxt:
            stage(0, 0, nxt)
        while nxt:
            slot = launch % self.depth
            cur, nxt = nxt, take()
            if nxt:                                    # copy of launch+1 overlaps compute of launch
                stage(launch + 1, (launch + 1) % self.depth, nxt)
            if len(pending) >= self.depth:             # the slot's host outputs must be free
                collect(*pending.pop(0))
            n = len(cur)
            with torch.cuda.stream(self.compute_stream):
                self.compute_stream.wait_event(self.copied[slot])
                if self.k == 1:
                    eng.detect(self.d_on[slot][0], self.fsmp, self.lsmp, self.available,
                               n_nodes_total=self.n_nodes_total,
                               out=tuple(t[0] for t in self.d_out[slot]))
                else:
                    eng.detect_batch(self.d_on[slot][:n], self.fsmp, self.lsmp, self.available,
                                     n_nodes_total=self.n_nodes_total,
                                     out=tuple(t[:n] for t in self.d_out[slot]))
                self.consumed[slot].record(self.compute_stream)
                for h, d in zip(self.h_out[slot], self.d_out[slot]):
                    h[:n].copy_(d[:n], non_blocking=True)
                self.done[slot].record(self.compute_stream)
            pending.append((step, slot, n))
            step += n
            launch += 1
        while pending:
            collect(*pending.pop(0))
        eng.set_stream(None)
        return results if on_result is None else step
