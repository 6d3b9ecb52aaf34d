#!/usr/bin/env python3
"""Randomised differential check of the native stream pipeline (qm_stream_*, csrc/qm_stream.hip; development aid).

Random small grids and tables, K = 1..5 timesteps per launch, depth 2..5, the three "stream_pull" settings (inputs
pulled by a kernel on the engine's stream / copied on a copy stream), random numbers of windows, and a random
interleaving of push / pop / flush the way a caller may drive it (pop whenever something is ready, flush in the
middle, a ring that runs full): every popped timestep against Engine.detect of the same window, bit for bit, in push
order.  usage: fuzz_stream.py [trials] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quakemigrate_amd.core import lib  # noqa: E402
from quakemigrate_amd.stream import StreamingDetector  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
pulled = copied = 0
for trial in range(trials):
    grid = tuple(int(v) for v in rng.integers(2, 20, size=3))
    S = int(rng.integers(1, 40))
    ns = int(rng.integers(1, 700))
    fsmp, lsmp = int(rng.integers(0, 20)), int(rng.integers(30, 120))
    ijk = np.stack(np.indices(grid), axis=-1).astype(np.float64)
    tt = np.empty(grid + (S,), dtype=np.int32)
    for r in range(S):
        src = rng.uniform(-5, np.array(grid) + 5)
        d = np.sqrt(((ijk - src) ** 2).sum(-1)) * rng.uniform(0.2, 9.0)
        tt[..., r] = np.minimum(np.rint(d - d.min()), lsmp).astype(np.int32)
    T = fsmp + ns + lsmp
    avail = int(rng.integers(1, S + 1))
    n_win = int(rng.integers(1, 14))
    wins = [np.log(np.clip(rng.lognormal(0, 0.6, size=(S, T)), 0.01, None)) for _ in range(n_win)]
    K, depth, pull = int(rng.integers(1, 6)), int(rng.integers(2, 6)), int(rng.integers(-1, 2))
    eng = lib.Engine(0, stream_pull=pull)
    eng.load_lut(tt)
    want = [eng.detect(w, fsmp, lsmp, avail) for w in wins]
    sd = StreamingDetector(eng, S, T, fsmp, lsmp, avail, depth=depth, steps_per_launch=K)
    got = []
    nxt = 0
    while len(got) < n_win:
        ready, filling = sd.pending()
        moves = []
        if nxt < n_win:
            moves.append("push")
        if ready:
            moves.append("pop")
        if filling:
            moves.append("flush")
        m = rng.choice(moves)
        if m == "push":
            if sd.push(wins[nxt]):
                nxt += 1
            else:                                    # the ring is full: something must be popped first
                assert ready > 0, (trial, "a full ring with nothing launched")
                k = int(rng.integers(1, ready + 1))
                a, b, c = sd.pop(k)
                got += [(a[j], b[j], c[j]) for j in range(k)]
        elif m == "pop":
            k = int(rng.integers(1, ready + 1))
            a, b, c = sd.pop(k)
            got += [(a[j], b[j], c[j]) for j in range(k)]
        else:
            sd.flush()
        if nxt == n_win and not sd.pending()[0] and sd.pending()[1]:
            sd.flush()
    assert sd.pending() == (0, 0)
    for j, (g, w) in enumerate(zip(got, want)):
        assert all(np.array_equal(x, y) for x, y in zip(g, w)), (trial, "timestep", j, grid, S, ns, K, depth, pull)
    words = S * T * K * 8
    pulled += pull > 0 or (pull < 0 and words <= (1 << 20))
    copied += not (pull > 0 or (pull < 0 and words <= (1 << 20)))
    sd.close()
    eng.close()
print(f"{trials} trials ok; slots pulled by a kernel in {pulled}, copied on the copy stream in {copied}")
