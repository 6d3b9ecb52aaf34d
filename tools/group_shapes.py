#!/usr/bin/env python3
"""Shift-reuse on coarse grids: LDS operands per add for node-group shapes other than 2x2x2 (CPU only).

The shift-reuse loop reads, per (group of nodes, table row), ONE window of 4 + (largest - smallest delay)
samples per lane (whole quads, at least kShiftNqMin = 4 of them, at most kShiftNqMax = 6) and feeds the
group's nodes from it: 4 x quads operands for 4 x nodes adds.  The round-2 kernels read 1.0 operand per add.
For a table and a list of group shapes this prints: the share of (group, row) pairs whose window fits the
register window (<= 6 quads), and the operands fetched per add over the pairs that fit -- the quantity
that decides whether a shape can beat the round-2 kernels on an LDS-bound table.
usage: group_shapes.py [C2|C3|E1|E2|...]"""
import itertools
import sys
import pathlib

import numpy as np

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from quakemigrate_amd import synth  # noqa: E402

NQMIN, NQMAX = 4, 6
for name in sys.argv[1:] or ["C2"]:
    case = synth.make_case(name)
    tt = np.maximum(case.traveltimes, 0)
    nx, ny, nz, S = tt.shape
    print(f"{name}: grid {nx}x{ny}x{nz}, {S} rows")
    print("  shape   nodes  fits<=6quads  mean quads  operands/add (fitting pairs)  operands/add if every pair ran (nq uncapped)")
    for shape in [(2, 2, 2), (2, 2, 1), (2, 1, 2), (1, 2, 2), (2, 1, 1), (1, 2, 1), (1, 1, 2), (1, 1, 4), (1, 1, 8), (1, 2, 4)]:
        gx, gy, gz = shape
        cx, cy, cz = nx // gx, ny // gy, nz // gz
        v = tt[:cx * gx, :cy * gy, :cz * gz].reshape(cx, gx, cy, gy, cz, gz, S)
        lo = v.min(axis=(1, 3, 5)).astype(np.int64)
        hi = v.max(axis=(1, 3, 5)).astype(np.int64)
        # the brick-relative alignment of e0 is unknown here: the worst of the four alignments on average
        e0 = lo & ~3
        nq = np.maximum((hi - e0 + 4 + 3) // 4, 2)
        fits = nq <= NQMAX
        fetched = np.maximum(nq, NQMIN)
        nodes = gx * gy * gz
        print(f"  {gx}x{gy}x{gz}   {nodes:3d}    {fits.mean():8.3f}     {nq.mean():7.2f}      "
              f"{(fetched[fits].mean() / nodes) if fits.any() else float('nan'):10.3f}"
              f"                        {np.maximum(nq, NQMIN).mean() / nodes:10.3f}")
