# -*- coding: utf-8 -*-
"""
Generate tests/golden/*.npz from the REFERENCE itself (authoring container only).

How: builds nothing itself -- expects ``make -C oracle`` to have produced
``oracle/_ref/qmlib.so`` from the reference's two C files where they lie under
/root/reference -- then imports the reference's own binding
``/root/reference/quakemigrate/core/lib.py`` (three stub
modules stand in for the obspy/pyproj-dependent package __init__ chain) and
records inputs and outputs of ``lib.migrate`` / ``lib.find_max_coa`` /
``lib.*_sta_lta``.  The fixtures are data only; no reference source travels.

Run:  python oracle/make_golden.py       (writes tests/golden/)
"""

import ctypes
import hashlib
import importlib.util
import pathlib
import platform
import sys
import types

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
REF = pathlib.Path("/root/reference")
REFLIB = ROOT / "oracle" / "_ref" / "qmlib.so"
OUT = ROOT / "tests" / "golden"


def load_reference_binding():
    qm = types.ModuleType("quakemigrate")
    qm.__path__ = []
    core = types.ModuleType("quakemigrate.core")
    core.__path__ = []
    util = types.ModuleType("quakemigrate.util")
    util.timeit = lambda *a, **k: (lambda f: f)
    ln = types.ModuleType("quakemigrate.core.libnames")
    ln._load_cdll = lambda name: ctypes.CDLL(str(REFLIB))
    sys.modules.update({"quakemigrate": qm, "quakemigrate.core": core,
                        "quakemigrate.util": util,
                        "quakemigrate.core.libnames": ln})
    qm.util = util
    spec = importlib.util.spec_from_file_location(
        "quakemigrate.core.lib", REF / "quakemigrate" / "core" / "lib.py")
    lib = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lib)
    return lib


def load_reference_lut_module():
    """
    The reference's quakemigrate/lut/lut.py, imported as a plain module.  It only fails to import
    here because pyproj is absent; LUT.serve_traveltimes and Grid3D.decimate, the two methods
    recorded below, never touch it, so an empty stand-in for the import is enough.
    """
    if "pyproj" not in sys.modules:
        pj = types.ModuleType("pyproj")
        pj.Transformer = type("Transformer", (), {})
        sys.modules["pyproj"] = pj
    spec = importlib.util.spec_from_file_location(
        "qm_reference_lut", REF / "quakemigrate" / "lut" / "lut.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def meta():
    return dict(generator="oracle/make_golden.py",
                reference="QuakeMigrate v1.2.1 core/lib.py + core/src/*.c",
                flags="-shared -fopenmp -fPIC -Ofast -lm -lgomp",
                gcc=platform.python_compiler(),
                glibc=" ".join(platform.libc_ver()),
                numpy=np.__version__)


def run(lib, onsets, tt, fsmp, lsmp, avail, threads=4):
    m = lib.migrate(onsets, tt, fsmp, lsmp, avail, threads)
    a, b, c = lib.find_max_coa(m, threads)
    return m, a, b, c


def save(name, **arrays):
    arrays["meta"] = np.array(repr(meta()))
    path = OUT / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"{name}: {path.stat().st_size / 1024:.0f} KiB")


def sample_rows(m, k=24, seed=5):
    vol = m.reshape(-1, m.shape[-1])
    rows = np.random.default_rng(seed).choice(vol.shape[0], size=min(k, vol.shape[0]),
                                              replace=False)
    rows.sort()
    return rows.astype(np.int64), vol[rows].copy()


def main():
    from quakemigrate_amd import synth

    OUT.mkdir(parents=True, exist_ok=True)
    lib = load_reference_binding()

    # 1. small random case, whole volume kept -------------------------------
    rng = np.random.default_rng(101)
    grid, S, ns, fsmp, lsmp = (9, 8, 7), 6, 150, 11, 47
    tt = rng.integers(0, lsmp + 1, size=grid + (S,), dtype=np.int32)
    on = np.clip(rng.lognormal(0, 0.6, size=(S, fsmp + ns + lsmp)), 0.4, None)
    m, a, b, c = run(lib, on, tt, fsmp, lsmp, S)
    save("small_random", onsets=on, traveltimes=tt, fsmp=fsmp, lsmp=lsmp,
         available=S, map4d=m, max_coa=a, max_norm_coa=b, max_coa_idx=c)

    # 2. ties: every onset on the clip floor -> every node equal -> index 0 ---
    grid, S, ns, fsmp, lsmp = (5, 4, 3), 4, 64, 3, 9
    tt = rng.integers(0, lsmp + 1, size=grid + (S,), dtype=np.int32)
    on = np.full((S, fsmp + ns + lsmp), 0.4)
    m, a, b, c = run(lib, on, tt, fsmp, lsmp, S)
    assert (c == 0).all()
    save("ties_floor", onsets=on, traveltimes=tt, fsmp=fsmp, lsmp=lsmp,
         available=S, map4d=m, max_coa=a, max_norm_coa=b, max_coa_idx=c)

    # 3. ties between two distant nodes with identical delay rows ------------
    grid, S, ns, fsmp, lsmp = (6, 5, 4), 5, 96, 7, 30
    tt = rng.integers(0, lsmp + 1, size=grid + (S,), dtype=np.int32)
    flat = tt.reshape(-1, S)
    lo, hi = 17, 101
    flat[hi] = flat[lo]                      # same delays -> same sums
    on = np.clip(rng.lognormal(0, 0.3, size=(S, fsmp + ns + lsmp)), 0.4, None)
    for r in range(S):                        # an event that both nodes see
        on[r, fsmp + 40 + flat[lo, r]] += 25.0
    m, a, b, c = run(lib, on, tt, fsmp, lsmp, S)
    assert c[40] == lo
    save("ties_twins", onsets=on, traveltimes=tt, fsmp=fsmp, lsmp=lsmp,
         available=S, map4d=m, max_coa=a, max_norm_coa=b, max_coa_idx=c,
         twin_lo=lo, twin_hi=hi)

    # 4. edges: negative delays clamp to 0, available != rows, fsmp = 0,
    #    lsmp == max delay exactly, onsets below the 0.01 clip and huge -------
    grid, S, ns, fsmp, lsmp = (4, 3, 5), 5, 80, 0, 21
    tt = rng.integers(-6, lsmp + 1, size=grid + (S,), dtype=np.int32)
    tt.reshape(-1, S)[7, 2] = lsmp
    on = np.clip(rng.lognormal(0, 1.0, size=(S, fsmp + ns + lsmp)), 0.0, None)
    on[0, ::7] = 0.0005
    on[1, 5::11] = 0.0
    on[3, 3::13] = 4.0e6
    m, a, b, c = run(lib, on, tt, fsmp, lsmp, S - 2)
    save("edges", onsets=on, traveltimes=tt, fsmp=fsmp, lsmp=lsmp,
         available=S - 2, map4d=m, max_coa=a, max_norm_coa=b, max_coa_idx=c)

    # 5. ragged sizes (nothing a multiple of any tile), volume sampled -------
    grid, S, ns, fsmp, lsmp = (23, 17, 13), 7, 333, 19, 110
    tt = np.ascontiguousarray(
        synth.homogeneous_lut(grid, 1.0, synth.station_positions(rng, grid, 1.0, S),
                              [5.0, 5.0, 5.0, 5.0, 2.9, 2.9, 2.9], 10.0))
    assert tt.max() <= lsmp
    on = np.clip(rng.lognormal(0, 0.5, size=(S, fsmp + ns + lsmp)), 0.4, None)
    m, a, b, c = run(lib, on, tt, fsmp, lsmp, S)
    rows, vals = sample_rows(m)
    save("ragged", onsets=on, traveltimes=tt, fsmp=fsmp, lsmp=lsmp, available=S,
         max_coa=a, max_norm_coa=b, max_coa_idx=c, map4d_rows=rows,
         map4d_vals=vals)

    # 6. Icequake_Iceland-sized geometry (BASELINE configs[0], C1), one step --
    case = synth.make_case("C1", step=0)
    m, a, b, c = run(lib, case.onsets, case.traveltimes, case.fsmp, case.lsmp,
                     case.available, threads=8)
    rows, vals = sample_rows(m)
    save("c1_icequake_geometry", onsets=case.onsets, lut_sha256=sha(case.traveltimes),
         fsmp=case.fsmp, lsmp=case.lsmp, available=case.available,
         max_coa=a, max_norm_coa=b, max_coa_idx=c, map4d_rows=rows,
         map4d_vals=vals, event_nodes=np.array(
             [np.ravel_multi_index(n[0], case.grid) for n in case.event_nodes]))
    del m

    # 7. shrunken C2 recipe (same generator the bench uses) ------------------
    case = synth.make_case("C2", step=0, grid=(26, 25, 14), n_samples=700)
    m, a, b, c = run(lib, case.onsets, case.traveltimes, case.fsmp, case.lsmp,
                     case.available, threads=8)
    rows, vals = sample_rows(m)
    save("c2_mini", onsets=case.onsets, lut_sha256=sha(case.traveltimes),
         grid=np.array(case.grid), fsmp=case.fsmp, lsmp=case.lsmp,
         available=case.available, max_coa=a, max_norm_coa=b, max_coa_idx=c,
         map4d_rows=rows, map4d_vals=vals)
    # quiet variant: whole step on the clip floor -> index 0 everywhere
    case = synth.make_case("C2", step=1, grid=(26, 25, 14), n_samples=200,
                           quiet=True)
    m, a, b, c = run(lib, case.onsets, case.traveltimes, case.fsmp, case.lsmp,
                     case.available, threads=8)
    assert (c == 0).all()
    save("c2_mini_quiet", lut_sha256=sha(case.traveltimes), grid=np.array(case.grid),
         fsmp=case.fsmp, lsmp=case.lsmp, available=case.available,
         t_samples=case.onsets.shape[1], max_coa=a, max_norm_coa=b, max_coa_idx=c)

    # 9. table serving + decimation through the reference's own LUT class ------
    lutmod = load_reference_lut_module()
    rng9 = np.random.default_rng(909)
    shape = (13, 12, 10)
    stations, phases = ["AAA", "BBB", "CCC", "DDD"], ["P", "S"]
    lut = lutmod.LUT.__new__(lutmod.LUT)              # bypass the pyproj-based constructor
    lut.node_count = np.array(shape, dtype=float)
    lut.node_spacing = np.array([0.5, 0.5, 0.5])
    lut.phases = phases
    lut.traveltimes = {}
    grids = {}
    for st in stations:
        for ph in phases:
            g = rng9.uniform(0.0, 9.0, size=shape)
            g[rng9.integers(0, shape[0]), rng9.integers(0, shape[1]), rng9.integers(0, shape[2])] = 2.5 / 50
            lut.traveltimes.setdefault(st, {})[ph] = g      # what LUT.__getitem__ serves
            grids[f"{st}_{ph}"] = g
    availability = {f"{st}_{ph}": 1 for ph in phases for st in stations}
    availability["CCC_P"] = 0
    availability["AAA_S"] = 0
    served = lut.serve_traveltimes(50, availability)
    dec = lut.decimate([2, 3, 4])
    served_dec = dec.serve_traveltimes(250, availability)
    save("serve_traveltimes",
         keys=np.array(list(grids.keys())), grids=np.stack(list(grids.values())),
         availability_keys=np.array(list(availability.keys())),
         availability_values=np.array(list(availability.values())),
         served_50=served, decimate=np.array([2, 3, 4]), served_dec_250=served_dec,
         dec_node_count=np.array(dec.node_count))

    # 10. onset stage: reference C STA/LTA (through lib.py) inside the NumPy glue of
    #     STALTAOnset._onset (stalta.py:515-546, :579-581) and lib.migrate (lib.py:93-94)
    from oracle import qm_oracle as oq

    rng10 = np.random.default_rng(1010)
    n_traces, T10 = 9, 1400
    trace_row = np.array([0, 0, 0, 1, 2, 2, 3, 3, 3], dtype=np.int32)    # 3-, 1-, 2-, 3-component rows
    nsta10 = np.array([11, 11, 21, 21], dtype=np.int32)
    nlta10 = np.array([51, 51, 101, 101], dtype=np.int32)
    sig10 = rng10.standard_normal((n_traces, T10)) * np.exp(rng10.normal(0, 1, (n_traces, 1)))
    sig10[:, 700:720] *= 12.0                         # an arrival
    sig10[4, 200:260] = 0.0                           # a dead stretch (lta -> 0 guard)
    arrays = dict(signals=sig10, trace_row=trace_row, nsta=nsta10, nlta=nlta10,
                  taper_pad=37, min_onset_value=0.4)
    for pos in ("classic", "centred"):
        for tf in ("energy", "abs"):
            raw, logged = oq.np_onset_stage(
                sig10, trace_row, nsta10, nlta10, tf, pos, 37, 0.4,
                stalta=(lib.overlapping_sta_lta, lib.centred_sta_lta))
            arrays[f"raw_{pos}_{tf}"] = raw
            arrays[f"log_{pos}_{tf}"] = logged
    raw, logged = oq.np_onset_stage(sig10, trace_row, nsta10, nlta10, "energy", "classic", -1,
                                    0.01, stalta=(lib.overlapping_sta_lta, lib.centred_sta_lta))
    arrays["raw_classic_energy_notaper"] = raw
    save("onset_stage", **arrays)

    # 8. STA/LTA: the reference's own known answers + a random trace ---------
    toy = np.arange(6)
    sig = np.abs(rng.standard_normal(600)) + 0.05
    save("stalta",
         toy=toy.astype(np.float64),
         toy_overlapping=lib.overlapping_sta_lta(toy, 2, 3),
         toy_centred=lib.centred_sta_lta(toy, 2, 3),
         toy_recursive=lib.recursive_sta_lta(toy, 2, 3),
         signal=sig, nsta=10, nlta=50,
         overlapping=lib.overlapping_sta_lta(sig, 10, 50),
         centred=lib.centred_sta_lta(sig, 10, 50),
         recursive=lib.recursive_sta_lta(sig, 10, 50))


if __name__ == "__main__":
    main()
