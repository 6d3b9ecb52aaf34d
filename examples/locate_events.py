# -*- coding: utf-8 -*-
"""
The reference's locate loop on the GPU engine, up to and including the location fits.

``QuakeScan._locate_events`` (quakemigrate/signal/scan.py:472-545) reads ``trigger_time -/+ (2 *
marginal_window + pad)`` per triggered event, computes the onsets and the 4-D coalescence map, takes the
time of the maximum as the origin time, drops the event if its trigger time is not inside ``origin -/+
marginal_window``, trims the map to that window, marginalises it over time and fits three locations.
``MigrationScan(stage="locate").locate_compute`` is that loop with the same plugin objects and the same
rules -- without the 4-D map: one fused detect launch over the event's window fixes the origin time, one
marginal-map launch sums the window inside the stacking kernel.  Picking, magnitudes and the event files
remain the reference's (``on_event`` is where they would be called).  Everything here is synthetic and
obspy-free.

Run:  python examples/locate_events.py
"""

import datetime as dt
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from quakemigrate_amd import scan, synth  # noqa: E402


def run(grid=(36, 32, 20), rows=12, rate=50, marginal_window=1.0, n_events=5):
    """``n_events`` synthetic events, one per trigger, each somewhere inside a 12-s record; the trigger
    times are what a detect run would hand over: within a fraction of the marginal window of the truth.
    (The third record ends too early for its event's window: the archive raises, the event is dropped.)"""
    mw, spacing = marginal_window, 0.5
    n_win = int(4 * mw * rate) + 1
    cases = [synth.make_case("C3", step=10 + s, grid=grid, rows=rows, n_samples=600, n_events=1,
                             table=(s == 0)) for s in range(n_events)]
    c0 = cases[0]
    keys = [f"ST{i % (rows // 2)}_{'P' if i < rows // 2 else 'S'}" for i in range(rows)]
    pre, post = c0.fsmp / rate, c0.lsmp / rate
    day = dt.datetime(2024, 5, 17, 10, 0, 0)
    # the event of record s has its origin at scanned sample t0 of that record; the trigger is 0.2 s late
    truth = [(c.event_nodes[0][0], c.event_nodes[0][1]) for c in cases]
    record_start = [day + dt.timedelta(seconds=60 * s) for s in range(n_events)]
    triggers = [(f"event_{s}", record_start[s] + dt.timedelta(seconds=truth[s][1] / rate + 0.2))
                for s in range(n_events)]

    record_of = {t - dt.timedelta(seconds=2 * mw + pre): s for s, (_, t) in enumerate(triggers)}

    class Data:
        pass

    class Archive:                                          # what the reference's Archive does for a window
        def read_waveform_data(self, w_beg, w_end):
            s = record_of[w_beg]
            first = int(round(((w_beg - record_start[s]).total_seconds() + pre) * rate))   # scanned sample
            n = int(round((w_end - w_beg).total_seconds() * rate)) + 1
            if first < 0 or first + n > cases[s].onsets.shape[1]:
                raise scan.DataGapException(f"the record of {triggers[s][0]} does not cover its window")
            d = Data()
            d.onsets, d.starttime = cases[s].onsets[:, first:first + n], w_beg
            return d

    class OnsetData:
        sampling_rate = rate
        availability = dict.fromkeys(keys, 1)

    class Onset:
        def calculate_onsets(self, data):
            return data.onsets, OnsetData()

    class Lut:
        node_spacing = np.array([spacing] * 3)

        def serve_traveltimes(self, sampling_rate, availability):
            return c0.traveltimes

        def index2coord(self, idx, unravel=True):
            return np.stack(np.unravel_index(idx, grid), axis=-1) * spacing

    s = scan.MigrationScan(Lut(), Onset(), pre, post, stage="locate", scan_rate=rate)
    located = s.locate_compute(Archive(), triggers, mw)
    assert all(r["max_coa"].shape[0] <= n_win for r in located)
    return located, truth, record_start, rate


if __name__ == "__main__":
    located, truth, record_start, rate = run()
    for r in located:
        k = int(r["uid"].split("_")[1])
        (node, t0), start = truth[k], record_start[k]
        origin = (r["otime"] - start).total_seconds() * rate
        print(f"{r['uid']}: origin sample {origin:.0f} (truth {t0}), spline node {np.round(r['fits'].spline, 2)}, "
              f"gaussian {np.round(r['fits'].gaussian, 2)} (truth {node}), "
              f"marginal window {r['last_sample'] - r['first_sample']} samples")
