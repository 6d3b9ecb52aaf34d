#!/usr/bin/env python3
"""Near-ties: how much of the reference's arg-max on near-tied maxima is its libm's rounding?

The reference compares EXPONENTIATED doubles with a strict '>' (migratelib.c:98-105); two nodes whose
float64 stacks differ by an ulp or two can round to the same exp() value (the lower index wins) or
swap order, so the index it returns on a near-tie depends on the exp() implementation the compiler
picked.  This script (CPU only, build container: it compiles the reference's own two C files where
they lie) runs the SAME inputs -- tests/golden/permuted_twins.npz and a mirror-twin family built
here -- through three builds of the reference's loops:

    vec     gcc -Ofast                      (the reference's flags: libmvec _ZGVbN2v_exp, 2 lanes)
    scalar  gcc -Ofast -fno-tree-vectorize  (glibc's scalar exp, correctly rounded in nearly all cases)
    O2      gcc -O2                         (scalar exp of stack / available, a true division)

and beside them the two rules an implementation can follow without the host's libm:

    sum     largest float64 stack, lowest index among equal stacks   (this engine)
    exp_cr  largest exp rounded from 80-bit arithmetic (<= 0.5 ulp + 2^-11), lowest index among equals

and prints the fraction of samples on which each pair disagrees.  usage: python tools/near_tie_study.py
"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/quakemigrate/core/src"


def build(tmp):
    libs = {}
    for name, flags in (("vec", ["-Ofast"]), ("scalar", ["-Ofast", "-fno-tree-vectorize"]),
                        ("O2", ["-O2"])):
        so = os.path.join(tmp, f"ref_{name}.so")
        subprocess.check_call(["gcc", "-shared", "-fopenmp", "-fPIC", *flags, f"-I{REF}",
                               f"{REF}/migratelib.c", f"{REF}/onsetlib.c", "-o", so, "-lm", "-lgomp"])
        libs[name] = ctypes.CDLL(so)
    return libs


def run(lib, lon, tt, fsmp, lsmp, avail):
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
    return c


def stacks(lon, tt, fsmp, lsmp):
    S, T = lon.shape
    ns = T - fsmp - lsmp
    flat = np.maximum(tt.reshape(-1, S), 0)
    out = np.zeros((flat.shape[0], ns))
    for r in range(S):                                  # ascending row order: the reference's sums
        idx = flat[:, r][:, None] + fsmp + np.arange(ns)[None, :]
        out += lon[r][idx]
    return out


def rules(lon, tt, fsmp, lsmp, avail):
    st = stacks(lon, tt, fsmp, lsmp)
    by_sum = st.argmax(axis=0)                          # first maximum
    z = (st * (1.0 / avail)).astype(np.longdouble)      # the argument the -Ofast builds form
    cr = np.exp(z).astype(np.float64)
    return by_sum, cr.argmax(axis=0), st


def mirror_family(rng, ns=512):
    """station pairs mirrored about the grid's mid-plane, seen with the same onset function: every
    sample's maximum is a near-tie between a node and its mirror image"""
    nx, ny, nz, half = 10, 8, 6, 5
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1) * 0.5
    st = rng.uniform([0, 0, 0], [4.5, 3.5, 0], size=(half, 3))
    mirror = st.copy()
    mirror[:, 0] = 4.5 - mirror[:, 0]
    xyz = np.concatenate([st, mirror])
    tt = np.stack([np.rint(np.sqrt(((g - p) ** 2).sum(-1)) / 3.0 * 50) for p in xyz], -1).astype(np.int32)
    lsmp = int(tt.max()) + 5
    rows = np.log(np.clip(rng.lognormal(0, 0.5, size=(half, 20 + ns + lsmp)), 0.4, np.inf))
    lon = np.ascontiguousarray(np.concatenate([rows, rows]))
    return lon, np.ascontiguousarray(tt), 20, lsmp, 2 * half


def report(name, lon, tt, fsmp, lsmp, avail, libs):
    got = {k: run(lib, lon, tt, fsmp, lsmp, avail) for k, lib in libs.items()}
    got["sum"], got["exp_cr"], st = rules(lon, tt, fsmp, lsmp, avail)
    keys = list(got)
    print(f"\n{name}: {len(got['sum'])} samples, {st.shape[0]} nodes; fraction of samples on which the "
          "arg-max differs")
    print("          " + "".join(f"{k:>9s}" for k in keys))
    for a in keys:
        print(f"{a:>9s} " + "".join(f"{float(np.mean(got[a] != got[b])):9.3f}" for b in keys))
    # whenever two answers differ: are the two nodes' stacks within a few ulp of each other?
    worst = 0.0
    for a in keys:
        for b in keys:
            d = np.flatnonzero(got[a] != got[b])
            if len(d):
                sa, sb = st[got[a][d], d], st[got[b][d], d]
                worst = max(worst, float(np.max(np.abs(sa - sb) / np.spacing(np.maximum(np.abs(sa), np.abs(sb))))))
    print(f"largest distance between the stacks of two disagreeing answers: {worst:.1f} ulp")
    return got


def main():
    if not os.path.exists(f"{REF}/migratelib.c"):
        sys.exit("the reference's sources are not present here")
    with tempfile.TemporaryDirectory() as tmp:
        libs = build(tmp)
        g = np.load(os.path.join(ROOT, "tests", "golden", "permuted_twins.npz"))
        lon = np.ascontiguousarray(np.log(np.clip(g["onsets"], 0.01, np.inf)))
        tt = np.ascontiguousarray(g["traveltimes"])
        got = report("permuted_twins.npz", lon, tt, int(g["fsmp"]), int(g["lsmp"]), int(g["available"]), libs)
        assert np.array_equal(got["vec"], g["max_coa_idx"]), "the fixture was made by the -Ofast build"
        report("mirror twins (5 station pairs)", *mirror_family(np.random.default_rng(7)), libs)
        print("\nglibc:", subprocess.check_output(["ldd", "--version"], text=True).splitlines()[0])


if __name__ == "__main__":
    main()
