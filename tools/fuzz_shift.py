#!/usr/bin/env python3
"""Randomised differential campaign for the shift-reuse kernel (development aid).

Random grids (every dimension >= 2), 1-64 rows (both workgroup shapes; a fifth of the trials 65-200 rows: row blocks
in any of their three forms), 1-900 scanned samples (round 4: every tail-tile length, with and without tail tiles), coherent tables of random
steepness (in 30 % of the trials with a few steep rows: bricks on the direct kernel, or a table that
does not qualify at all), quantised onsets in half of the trials (exact ties),  negative delays, random `available` and
group counts: the automatic engine against Engine(shift=0) (maxima bit for bit, indices) and the
oracle; every fourth trial also the volume-writing variant against the oracle's volume, every third
the marginalised map of a random window against the time sum of that volume, every fifth a batch of
two or three timesteps in one launch against the steps one by one, every fourth the opt-in tie_rule = 1 (alone and with two timesteps in one launch; every eighth on round 5's sets of bricks; every twelfth also sharded over two engines through qm_engine_tie_partial / _tie_fold; an eighth of the trials also on MIRROR TWINS made from the trial's table -- near-ties at every sample)
against the oracle's restatement of the reference's exp rule.  Round 6: three trials in ten scan 384-2100
samples, and half of all trials ask for the WIDE tiles (six samples per lane) wherever the scan holds one.
On a mismatch the trial's diagnosis is printed before the assertion (which kernel, which nodes and
samples, whether a second run moves): the round-4 GPU-sharing study started from these lines.
usage: fuzz_shift.py [trials] [seed]      (QM_FUZZ_ONLY=trial [QM_FUZZ_REPEAT=n]: that trial of the seed only)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import qm_oracle  # noqa: E402
from quakemigrate_amd.core import lib  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
used = wide = blocks = six = 0
repeat = int(os.environ.get("QM_FUZZ_REPEAT", "1"))    # (with QM_FUZZ_ONLY: the trial's engine sequence this many times)
for trial in range(trials):
    grid = tuple(int(v) for v in rng.integers(2, 34, size=3))
    if np.prod(grid) > 12000:
        grid = (grid[0], grid[1], max(2, 12000 // (grid[0] * grid[1])))
    S = int(rng.integers(1, 65)) if rng.random() < 0.8 else int(rng.integers(65, 201))   # (row blocks)
    ns = int(rng.integers(1, 900)) if rng.random() < 0.7 else int(rng.integers(192, 900))
    if rng.random() < 0.3:              # (round 6: scans that hold one to five WIDE tiles of 384 samples)
        ns = int(rng.integers(384, 2100))
    fsmp, lsmp = int(rng.integers(0, 30)), int(rng.integers(30, 160))
    # coherent table: distance-like delays from random "stations", steepness up to ~7 samples per node, a fifth of the rows up to 30
    ijk = np.stack(np.indices(grid), axis=-1).astype(np.float64)
    tt = np.empty(grid + (S,), dtype=np.int32)
    rough = rng.random() < 0.3          # (a steep row disqualifies the table or sends bricks to the direct kernel)
    for r in range(S):
        src = rng.uniform(-5, np.array(grid) + 5)
        steep = rng.uniform(0.2, 7.0) if (not rough or rng.random() < 0.9) else rng.uniform(7.0, 30.0)
        d = np.sqrt(((ijk - src) ** 2).sum(-1)) * steep
        tt[..., r] = np.minimum(np.rint(d - d.min() + rng.integers(0, 5)), lsmp).astype(np.int32)
    if trial % 3 == 0:
        tt[(tt <= 1) & (rng.random(tt.shape) < 0.5)] = -3          # clamp to 0 (migratelib.c:55)
    T = fsmp + ns + lsmp
    if trial % 2:
        lon = rng.choice([-48, -16, 0, 16, 80], size=(S, T), p=[0.45, 0.25, 0.15, 0.1, 0.05]) / 64.0
    else:
        lon = np.log(np.clip(rng.lognormal(0, 0.6, size=(S, T)), 0.01, None))
    avail = int(2 ** rng.integers(0, 5)) if trial % 2 else int(rng.integers(1, S + 1))
    cfg = dict(groups=int(rng.choice([0, 1, 3, 9])), shift_lazy=int(rng.integers(-1, 2)),
               shift=1 if 64 < S <= 96 else -1, shift_rows_direct=int(rng.integers(0, 3)),
               shift_tail=int(rng.integers(0, 2)),
               # (round 6: the fused detect on wide tiles where the table has that layout -- forced on small grids,
               # where the automatic rule would not take it)
               shift_wide=int(rng.choice([1, 1, 0, -1])),
               # (... on row blocks: beyond 64 rows by itself, in a third of the trials whatever the row count)
               shift_wide_rows=int(rng.choice([1, 1, 2])))
    if os.environ.get("QM_FUZZ_ONLY") and trial != int(os.environ["QM_FUZZ_ONLY"]):
        # (replay one trial of a seed: the random stream has to advance as in the full run, including
        # the draws the skipped checks would have made)
        if trial % 3 == 0:
            i0 = int(rng.integers(0, ns))
            int(rng.integers(i0 + 1, ns + 1))
        if trial % 5 == 0:
            int(rng.integers(2, 4))
        continue
    want = qm_oracle.detect(lon, tt, fsmp, lsmp, avail, threads=4, prelogged=True)
    if os.environ.get("QM_FUZZ_ONLY"):                              # ... with a diagnosis
        print("trial", trial, "grid", grid, "S", S, "ns", ns, "fsmp", fsmp, "lsmp", lsmp, "avail", avail, cfg)
        for direct in (1, 0):
            eng = lib.Engine(0, **{**cfg, "shift_rows_direct": direct})
            eng.load_lut(tt)
            got = eng.detect(lon, fsmp, lsmp, avail)
            bad = np.flatnonzero((got[2] != want[2]) | ~np.isclose(got[0], want[0], rtol=1e-12, atol=0))
            print(" direct", direct, "kernel", eng.get("last_kernel"), "blocks", eng.get("shift_row_blocks"),
                  "wide", eng.get("shift_wide_bricks"), "bad samples", bad.size, bad[:24],
                  "got idx", got[2][bad[:6]], "want idx", want[2][bad[:6]],
                  "got", got[0][bad[:4]], "want", want[0][bad[:4]])
            eng.close()
    if trial % 3 == 0:                                              # (the draws of the checks below, in their order)
        i0 = int(rng.integers(0, ns))
        i1 = int(rng.integers(i0 + 1, ns + 1))
    if trial % 5 == 0:
        k = int(rng.integers(2, 4))
    ref = None
    if trial % 4 == 0 or trial % 3 == 0:
        ref = qm_oracle.c_migrate(lon, tt, fsmp, lsmp, avail, threads=4, prelogged=True)
    for rep in range(repeat if os.environ.get("QM_FUZZ_ONLY") else 1):
        res = {}
        for tag, extra in (("shift", {}), ("round2", {"shift": 0, "shift_lazy": -1})):
            eng = lib.Engine(0, **{**cfg, **extra})
            eng.load_lut(tt)
            res[tag] = eng.detect(lon, fsmp, lsmp, avail)
            res_first = res[tag]
            if tag == "shift":
                kern, nwide = eng.get("last_kernel"), eng.get("shift_wide_bricks")
                six_now = kern == 3 and eng.get("shift_wide_tiles") > 0
                if trial % 4 == 0:
                    vol = np.zeros(grid + (ns,))
                    series = (np.zeros(ns), np.zeros(ns), np.zeros(ns, dtype=np.int64))
                    eng.migrate(lon, fsmp, lsmp, avail, vol, scan_out=series)
                    if not np.allclose(vol, ref, rtol=1e-13, atol=0):
                        bad = ~np.isclose(vol, ref, rtol=1e-13, atol=0)
                        fv, fb = vol.reshape(-1, ns), bad.reshape(-1, ns)
                        nodes = np.flatnonzero(fb.any(axis=1))
                        cols = np.flatnonzero(fb.any(axis=0))
                        print("VOLUME MISMATCH", trial, grid, S, ns, cfg, "kernel", eng.get("last_kernel"), "tail",
                              eng.get("shift_tail_spl"), "waves", eng.get("shift_waves"), "blocks", eng.get("shift_row_blocks"),
                              "wide", eng.get("shift_wide_bricks"), "\n nodes", len(nodes), nodes[:40], "\n samples", len(cols),
                              cols[:8], "..", cols[-8:], "\n got", fv[nodes[0], cols[:6]], "want", ref.reshape(-1, ns)[nodes[0], cols[:6]],
                              "\n node coords", [tuple(int(v) for v in np.unravel_index(n, grid)) for n in nodes[:24]], flush=True)
                        flat = np.flatnonzero(bad.ravel())
                        vol2 = np.zeros(grid + (ns,))
                        eng.migrate(lon, fsmp, lsmp, avail, vol2)
                        print("  flat elements", flat[0], "..", flat[-1], "count", flat.size, "byte offset of the first", flat[0] * 8,
                              "base address mod 4096", vol.ctypes.data % 4096, "runs", 1 + int((np.diff(flat) > 1).sum()),
                              "\n  a second migrate on the same engine matches the oracle:",
                              bool(np.allclose(vol2, ref, rtol=1e-13, atol=0)), flush=True)
                    np.testing.assert_allclose(vol, ref, rtol=1e-13, err_msg=str((trial, grid, S, ns)))
                    assert np.array_equal(series[2], want[2]), (trial, "volume scan idx")
                if trial % 3 == 0:
                    series = (np.zeros(ns), np.zeros(ns), np.zeros(ns, dtype=np.int64))
                    m = eng.marginal_map(lon, fsmp, lsmp, avail, i0, i1, scan_out=series)
                    np.testing.assert_allclose(m, ref[..., i0:i1].sum(axis=-1), rtol=2e-12,
                                               err_msg=str((trial, grid, S, ns, i0, i1, cfg)))
                    assert np.array_equal(series[2], want[2]) and np.array_equal(series[0], res_first[0]), \
                        (trial, "marginal scan", cfg)
                if trial % 4 == 0:                                 # (round 5) the opt-in exp rule on near-ties
                    # (round 6: the shift-reuse fused detect refines from a row of maxima per brick; every other
                    # tie trial on round 5's sets of bricks instead)
                    te = lib.Engine(0, tie_rule=1, tie_sets=int(trial % 8 != 0), **cfg)
                    te.load_lut(tt)
                    tied = te.detect(lon, fsmp, lsmp, avail)
                    clean = te.get("tie_overflow_samples") == 0
                    if clean:
                        rule = qm_oracle.np_argmax_exp_rule(lon, tt, fsmp, lsmp, avail, prelogged=True)
                        assert np.array_equal(tied[2], rule), (trial, "tie_rule", cfg, te.get("tie_brick_rows"),
                                                               np.flatnonzero(tied[2] != rule)[:8])
                    assert np.array_equal(tied[0], res_first[0]), (trial, "tie_rule values")
                    np.testing.assert_allclose(tied[1], res_first[1], rtol=1e-13)
                    if ns >= 8:
                        # ... and two timesteps in one launch keep the refinement, step for step
                        other = np.ascontiguousarray(lon[:, ::-1])
                        want_other = te.detect(other, fsmp, lsmp, avail)
                        both = te.detect_batch(np.stack([lon, other]), fsmp, lsmp, avail)
                        for j, one in enumerate((tied, want_other)):
                            for i in range(3):
                                if not np.array_equal(both[i][j], one[i]):
                                    bad = np.flatnonzero(both[i][j] != one[i])
                                    print("TIE BATCH MISMATCH", trial, grid, S, ns, cfg, "step", j, "array", i, "bad samples",
                                          len(bad), bad[:12], "batch", both[i][j][bad[:6]], "single", one[i][bad[:6]],
                                          "rows", te.get("tie_brick_rows"), "overflow", te.get("tie_overflow_samples"),
                                          "kernel", te.get("last_kernel"), te.get("last_kernel_j"), "wide tiles",
                                          te.get("shift_wide_tiles"), flush=True)
                        assert all(np.array_equal(both[i][0], tied[i]) and np.array_equal(both[i][1], want_other[i])
                                   for i in range(3)), (trial, "tie_rule in a batch", cfg, te.get("steps_per_launch"))
                    te.close()
                    if trial % 8 == 4 and S % 2 == 0 and S >= 2:
                        # (round 6) MIRROR TWINS: the second half of the rows are the first half's stations mirrored
                        # about the grid's mid x-plane, seen with the same log-onsets -- node (i, j, k) and node
                        # (nx - 1 - i, j, k) stack the same multiset in another row order: every sample's maximum
                        # is a NEAR-tie (0-2 ulps) between two nodes of different bricks.  The rule against the
                        # oracle's restatement; the values against the default engine on the same data.
                        tt_m, lon_m = tt.copy(), np.ascontiguousarray(np.log(np.clip(
                            np.random.default_rng(trial).lognormal(0, 0.6, size=(S, T)), 0.01, None)))
                        tt_m[..., S // 2:] = tt_m[::-1, ..., :S // 2]
                        lon_m[S // 2:] = lon_m[:S // 2]
                        dm = lib.Engine(0, **cfg)
                        dm.load_lut(tt_m)
                        base_m = dm.detect(lon_m, fsmp, lsmp, avail)
                        dm.close()
                        tm = lib.Engine(0, tie_rule=1, **cfg)
                        tm.load_lut(tt_m)
                        tied_m = tm.detect(lon_m, fsmp, lsmp, avail)
                        if tm.get("tie_overflow_samples") == 0:
                            rule_m = qm_oracle.np_argmax_exp_rule(lon_m, tt_m, fsmp, lsmp, avail, prelogged=True)
                            assert np.array_equal(tied_m[2], rule_m), (trial, "tie_rule on mirror twins", cfg, grid, S, ns,
                                                                       tm.get("tie_brick_rows"), tm.get("last_kernel"),
                                                                       np.flatnonzero(tied_m[2] != rule_m)[:8])
                        assert np.array_equal(tied_m[0], base_m[0]), (trial, "tie_rule on mirror twins: values")
                        np.testing.assert_allclose(tied_m[1], base_m[1], rtol=1e-13)
                        tm.close()
                    if trial % 12 == 0 and grid[0] >= 2:
                        # ... and sharded: the grid cut into two x-slabs on two engines, the exchange of
                        # distributed.ShardedDetector done by hand in this process (partials -> packed fold ->
                        # every engine's candidates against the GRID's maxima -> fold of the outcomes)
                        import torch
                        from quakemigrate_amd import distributed as qd
                        mx = grid[0] // 2
                        slabs = [(0, mx), (mx, grid[0])]
                        engs = []
                        for x0, x1 in slabs:
                            se = lib.Engine(0, tie_rule=1, **cfg)
                            se.set_stream(torch.cuda.current_stream().cuda_stream)    # (one stream orders the engines)
                            se.load_lut(np.ascontiguousarray(tt[x0:x1]), node_offset=x0 * grid[1] * grid[2])
                            engs.append(se)
                        dlon = torch.from_numpy(lon).cuda()
                        gathered = torch.empty((2, 3, ns), dtype=torch.float64, device="cuda")
                        for r, se in enumerate(engs):
                            se.detect_partial(dlon, fsmp, lsmp, avail, (gathered[r, 0], gathered[r, 1].view(torch.int64),
                                                                        gathered[r, 2]))
                        out = tuple(torch.empty(ns, dtype=d, device="cuda") for d in (torch.float64, torch.float64, torch.int64))
                        engs[0].finalize_packed(gathered, 2, ns, int(np.prod(grid)), out=out)
                        tgat = torch.zeros((2, 2, ns), dtype=torch.float64, device="cuda")
                        for r, se in enumerate(engs):
                            se.tie_partial(dlon, fsmp, lsmp, avail, gathered, 2, tgat[r])
                        engs[0].tie_fold(tgat, 2, ns, out[2])
                        torch.cuda.synchronize()
                        over = sum(se.get("tie_overflow_samples") for se in engs)
                        if over == 0 and clean:
                            got_idx = out[2].cpu().numpy()
                            assert np.array_equal(got_idx, rule), (trial, "sharded tie_rule", cfg, grid,
                                                                   np.flatnonzero(got_idx != rule)[:8])
                        assert np.array_equal(out[0].cpu().numpy(), res_first[0]), (trial, "sharded tie_rule values")
                        for se in engs:
                            se.close()
                if trial % 5 == 0:
                    lons = np.stack([lon] + [np.roll(lon, 7 * (j + 1), axis=1) for j in range(k - 1)])
                    both = eng.detect_batch(lons, fsmp, lsmp, avail)
                    for j in range(k):
                        one = eng.detect(lons[j], fsmp, lsmp, avail)
                        assert all(np.array_equal(both[i][j], one[i]) for i in range(3)), (trial, "batch", j, cfg)
            eng.close()
        used += kern == 3
        blocks += kern == 3 and S > 64
        wide += kern == 3 and nwide > 0
        six += six_now
        a, b, c = res["shift"]
        if not (np.array_equal(c, want[2]) and np.array_equal(c, res["round2"][2]) and np.array_equal(a, res["round2"][0])):
            badt = np.flatnonzero((c != want[2]) | (a != res["round2"][0]) | (c != res["round2"][2]))
            print("DETECT MISMATCH", trial, grid, S, ns, cfg, "kernel", kern, "bad samples", len(badt), badt[:16], "..", badt[-8:],
                  "\n got idx", c[badt[:8]], "want", want[2][badt[:8]], "\n got max", a[badt[:4]], "round2", res["round2"][0][badt[:4]],
                  "want", want[0][badt[:4]], "\n round2 idx", res["round2"][2][badt[:8]],
                  "round2 vs oracle: idx", np.array_equal(res["round2"][2], want[2]), "max", np.allclose(res["round2"][0], want[0], rtol=1e-13),
                  "shift vs oracle: max", np.allclose(a, want[0], rtol=1e-13), flush=True)
            for tag, extra in (("shift", {}), ("round2", {"shift": 0, "shift_lazy": -1})):     # which one moves on a second run?
                eng = lib.Engine(0, **{**cfg, **extra})
                eng.load_lut(tt)
                again = eng.detect(lon, fsmp, lsmp, avail)
                print("  again", tag, "same as before:", [bool(np.array_equal(again[i], res[tag][i])) for i in range(3)], flush=True)
                eng.close()
        assert np.array_equal(c, want[2]), (trial, grid, S, ns, cfg, kern)
        assert np.array_equal(c, res["round2"][2]) and np.array_equal(a, res["round2"][0]), (trial, grid, S, ns)
        np.testing.assert_allclose(a, want[0], rtol=1e-13)
        np.testing.assert_allclose(b, want[1], rtol=2e-12)   # degree-8 2^f: truncation 7.8e-13 + rounding
        np.testing.assert_allclose(b, res["round2"][1], rtol=2e-12)
print(f"{trials} trials ok; shift kernel used in {used} ({blocks} of them on row blocks, {six} on wide tiles), of "
      f"which {wide} with bricks on the direct kernel")
