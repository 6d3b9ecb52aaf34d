# -*- coding: utf-8 -*-
"""
The reference's detect loop on the GPU engine, plugin objects in, ``.scanmseed`` out.

``QuakeScan._continuous_compute`` (quakemigrate/signal/scan.py:407-470) reads a timestep's waveforms,
computes its onsets, migrates, scans and appends -- one timestep after the other, an all-zero timestep
where the archive has no data.  ``MigrationScan.continuous_compute`` is that loop with the same plugin
objects (``archive.read_waveform_data``, ``onset.calculate_onsets``, ``lut.serve_traveltimes`` /
``index2coord``) and the same behaviour around the path; the timesteps go through the library's native
pipeline (copies on their own HIP streams, several timesteps per launch), the results reach the sink
in order.  Everything here is synthetic and obspy-free: the three plugin classes below stand where the
reference's ``Archive``, ``STALTAOnset`` and ``LUT`` objects would.

Run:  python examples/continuous_detect.py [out_dir]
"""

import datetime as dt
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from quakemigrate_amd import scan, scanmseed, synth  # noqa: E402


def run(out_dir, n_steps=12, grid=(36, 32, 20), rows=12, rate=50, n_samples=600):
    cases = [synth.make_case("C3", step=s, grid=grid, rows=rows, n_samples=n_samples, table=(s == 0))
             for s in range(n_steps)]
    c0 = cases[0]
    keys = [f"ST{i % (rows // 2)}_{'P' if i < rows // 2 else 'S'}" for i in range(rows)]
    timestep, pre, post = n_samples / rate, c0.fsmp / rate, c0.lsmp / rate
    t0 = dt.datetime(2024, 5, 17, 23, 58, 0)               # (the run crosses midnight: two files)

    class Data:                                             # what archive.read_waveform_data returns
        def __init__(self, i, starttime):
            self.i, self.starttime = i, starttime

    class Archive:
        def read_waveform_data(self, w_beg, w_end):
            i = int(round(((w_beg - t0).total_seconds() + pre) / timestep))
            if i == 4:                                      # a gap in the archive
                raise scan.DataGapException(f"no data between {w_beg} and {w_end}")
            return Data(i, w_beg)

    class OnsetData:
        sampling_rate = rate
        availability = dict.fromkeys(keys, 1)

    class Onset:                                            # the onset plugin (base.py:103-106)
        def calculate_onsets(self, data):
            return cases[data.i].onsets, OnsetData()

    class Lut:                                              # the LUT plugin (lut.py:502-538, 211-243)
        unit_conversion_factor = 1000.0

        def serve_traveltimes(self, sampling_rate, availability):
            return c0.traveltimes

        def index2coord(self, idx, unravel=True):
            return np.stack(np.unravel_index(idx, grid), axis=-1) * 0.5

    sink = scanmseed.CoalescenceSink(out_dir, rate)
    s = scan.MigrationScan(Lut(), Onset(), pre, post, stage="detect")
    availability = s.continuous_compute(Archive(), t0, n_steps, timestep, rate, sink)
    return sink, availability, cases


if __name__ == "__main__":
    sink, availability, cases = run(pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "continuous_detect_out"))
    for path in sink.files:
        start, rate, cols = scanmseed.read_scanmseed(path, ucf=1000.0)
        print(f"{path.name}: {len(cols['COA'])} samples from {start}, peak coalescence {cols['COA'].max():.2f}")
    print("timesteps without data:", [i for i, row in enumerate(availability) if not any(row.values())])
