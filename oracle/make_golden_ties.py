# -*- coding: utf-8 -*-
"""
Generate tests/golden/near_ties_scalar.npz from the REFERENCE itself (authoring container only).

The reference compares exponentiated doubles (migratelib.c:98-105), so on near-tied maxima its
arg-max depends on the exp() it was compiled against.  ``make -C oracle ref`` builds the reference's
two C files twice where they lie: with its own flags (``_ref/qmlib.so``: -Ofast, libmvec's two-lane
exp) and with ``-fno-tree-vectorize`` added (``_ref/qmlib_scalar.so``: glibc's scalar exp, correctly
rounded in all but ~0.07 % of its arguments).  This script runs both on the two adversarial families
-- tests/golden/permuted_twins.npz (inputs already a fixture) and a mirror-twin family (built here,
inputs stored) -- and records the index series.  The engine's opt-in ``tie_rule = 1`` is pinned on the
scalar build's (tests/test_gpu_parity.py).  Data only; no reference source travels.

Run:  make -C oracle ref && python oracle/make_golden_ties.py
"""
import ctypes
import pathlib
import platform

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "golden"


def run(lib, lon, tt, fsmp, lsmp, avail):
    """migrate + find_max_coa through the reference's C symbols (qmlib.h:28-32), log-onsets in."""
    S, T = lon.shape
    ns = T - fsmp - lsmp
    n = int(np.prod(tt.shape[:-1]))
    vol = np.zeros((n, ns))
    P = ctypes.c_void_p
    lib.migrate(P(lon.ctypes.data), P(tt.ctypes.data), P(vol.ctypes.data), ctypes.c_int32(fsmp),
                ctypes.c_int32(lsmp), ctypes.c_int32(ns), ctypes.c_int32(S), ctypes.c_int32(avail),
                ctypes.c_int64(n), ctypes.c_int64(1))
    a, b, c = np.zeros(ns), np.zeros(ns), np.zeros(ns, dtype=np.int64)
    lib.find_max_coa(P(vol.ctypes.data), P(a.ctypes.data), P(b.ctypes.data), P(c.ctypes.data),
                     ctypes.c_int32(ns), ctypes.c_int64(n), ctypes.c_int64(1))
    return a, b, c


def mirror_family(rng, ns=512, shape=(10, 8, 6)):
    """station pairs mirrored about the grid's mid-plane, seen with the same onset function: every
    sample's maximum is a near-tie between a node and its mirror image (tools/near_tie_study.py)"""
    (nx, ny, nz), half = shape, 5
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1) * 0.5
    span = 0.5 * (nx - 1)
    st = rng.uniform([0, 0, 0], [span, 0.5 * (ny - 1), 0], size=(half, 3))
    mirror = st.copy()
    mirror[:, 0] = span - mirror[:, 0]
    xyz = np.concatenate([st, mirror])
    tt = np.stack([np.rint(np.sqrt(((g - p) ** 2).sum(-1)) / 3.0 * 50) for p in xyz], -1).astype(np.int32)
    lsmp = int(tt.max()) + 5
    rows = np.clip(rng.lognormal(0, 0.5, size=(half, 20 + ns + lsmp)), 0.4, np.inf)
    return np.ascontiguousarray(np.concatenate([rows, rows])), np.ascontiguousarray(tt), 20, lsmp, 2 * half


def main():
    vec = ctypes.CDLL(str(ROOT / "oracle" / "_ref" / "qmlib.so"))
    scalar = ctypes.CDLL(str(ROOT / "oracle" / "_ref" / "qmlib_scalar.so"))
    g = np.load(OUT / "permuted_twins.npz")
    lon = np.ascontiguousarray(np.log(np.clip(g["onsets"], 0.01, np.inf)))
    tt = np.ascontiguousarray(g["traveltimes"])
    args = (int(g["fsmp"]), int(g["lsmp"]), int(g["available"]))
    pv, ps = run(vec, lon, tt, *args), run(scalar, lon, tt, *args)
    assert np.array_equal(pv[2], g["max_coa_idx"]), "permuted_twins.npz was made by the -Ofast build"
    on, mtt, fsmp, lsmp, avail = mirror_family(np.random.default_rng(7))
    mlon = np.ascontiguousarray(np.log(np.clip(on, 0.01, np.inf)))
    mv, ms = run(vec, mlon, mtt, fsmp, lsmp, avail), run(scalar, mlon, mtt, fsmp, lsmp, avail)
    meta = (f"made by oracle/make_golden_ties.py from the reference's migratelib.c / onsetlib.c, gcc "
            f"-Ofast [-fno-tree-vectorize], glibc {platform.libc_ver()[1]}, numpy {np.__version__}")
    np.savez_compressed(
        OUT / "near_ties_scalar.npz", meta=np.array(meta),
        permuted_idx_scalar=ps[2], permuted_max_coa_scalar=ps[0], permuted_max_norm_coa_scalar=ps[1],
        mirror_onsets=on, mirror_traveltimes=mtt, mirror_fsmp=fsmp, mirror_lsmp=lsmp, mirror_available=avail,
        mirror_idx_vec=mv[2], mirror_idx_scalar=ms[2], mirror_max_coa_scalar=ms[0],
        mirror_max_norm_coa_scalar=ms[1])
    print("permuted twins: scalar vs -Ofast differ on", float(np.mean(pv[2] != ps[2])),
          "; mirror twins:", float(np.mean(mv[2] != ms[2])))
    # Round 6: the same family on a grid of many bricks (a node and its mirror image lie in DIFFERENT bricks, 45 of
    # 8x8x8 / 30 of 8x8x16 nodes) and a scan that holds wide tiles: what the per-brick partial sets of tie_rule = 1
    # and its sharded form are pinned on (near_ties_bricks.npz)
    on, btt, fsmp, lsmp, avail = mirror_family(np.random.default_rng(11), ns=1600, shape=(40, 24, 20))
    blon = np.ascontiguousarray(np.log(np.clip(on, 0.01, np.inf)))
    bv, bs = run(vec, blon, btt, fsmp, lsmp, avail), run(scalar, blon, btt, fsmp, lsmp, avail)
    np.savez_compressed(
        OUT / "near_ties_bricks.npz", meta=np.array(meta), onsets=on, traveltimes=btt, fsmp=fsmp, lsmp=lsmp,
        available=avail, idx_vec=bv[2], idx_scalar=bs[2], max_coa_scalar=bs[0], max_norm_coa_scalar=bs[1])
    print("mirror twins on (40, 24, 20): scalar vs -Ofast differ on", float(np.mean(bv[2] != bs[2])))


if __name__ == "__main__":
    main()
