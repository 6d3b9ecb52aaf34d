# -*- coding: utf-8 -*-
"""
Pin the CPU oracle (oracle/qm_oracle.c + the NumPy restatement) to the golden
vectors recorded from the reference itself (oracle/make_golden.py), and to the
reference's own known answers for the STA/LTA functions
(/root/reference tests/test_onsets.py:27-35).  CPU only.
"""

import hashlib

import numpy as np
import pytest

from conftest import RTOL, load_golden
from quakemigrate_amd import synth

FULL = ["small_random", "ties_floor", "ties_twins", "edges"]


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", FULL)
def test_c_port_matches_reference_full_volume(oracle, name):
    g = load_golden(name)
    m = oracle.c_migrate(g["onsets"], g["traveltimes"], int(g["fsmp"]),
                         int(g["lsmp"]), int(g["available"]), threads=3)
    # same compiler, same flags, same loop nest -> identical bits expected;
    # the contract is 1e-6 relative.
    np.testing.assert_allclose(m, g["map4d"], rtol=1e-14, atol=0)
    a, b, c = oracle.c_find_max_coa(m, threads=2)
    assert np.array_equal(c, g["max_coa_idx"])
    np.testing.assert_allclose(a, g["max_coa"], rtol=1e-14)
    np.testing.assert_allclose(b, g["max_norm_coa"], rtol=1e-14)


@pytest.mark.parametrize("name", FULL)
def test_numpy_restatement_matches_reference(oracle, name):
    g = load_golden(name)
    m = oracle.np_migrate(g["onsets"], g["traveltimes"], int(g["fsmp"]),
                          int(g["lsmp"]), int(g["available"]))
    np.testing.assert_allclose(m, g["map4d"], rtol=1e-13, atol=0)
    # scan the REFERENCE volume so the index comparison is exact
    a, b, c = oracle.np_find_max_coa(g["map4d"])
    assert np.array_equal(c, g["max_coa_idx"])
    np.testing.assert_allclose(a, g["max_coa"], rtol=0)
    np.testing.assert_allclose(b, g["max_norm_coa"], rtol=1e-13)


def test_ties_resolve_to_lowest_index(oracle):
    g = load_golden("ties_floor")
    assert (g["max_coa_idx"] == 0).all()
    g = load_golden("ties_twins")
    lo, hi = int(g["twin_lo"]), int(g["twin_hi"])
    vol = g["map4d"].reshape(-1, g["map4d"].shape[-1])
    assert np.array_equal(vol[lo], vol[hi])
    assert g["max_coa_idx"][40] == lo


def test_ragged_sampled_volume_and_chunked_detect(oracle):
    g = load_golden("ragged")
    m = oracle.c_migrate(g["onsets"], g["traveltimes"], int(g["fsmp"]),
                         int(g["lsmp"]), int(g["available"]), threads=4)
    vol = m.reshape(-1, m.shape[-1])
    np.testing.assert_allclose(vol[g["map4d_rows"]], g["map4d_vals"], rtol=1e-14)
    # time-chunked detect == one-shot reference outputs
    a, b, c = oracle.detect(g["onsets"], g["traveltimes"], int(g["fsmp"]),
                            int(g["lsmp"]), int(g["available"]), threads=4,
                            max_bytes=8 * vol.shape[0] * 50)
    assert np.array_equal(c, g["max_coa_idx"])
    np.testing.assert_allclose(a, g["max_coa"], rtol=1e-14)
    np.testing.assert_allclose(b, g["max_norm_coa"], rtol=1e-14)


def test_c2_mini_recipe_is_reproducible_and_matches(oracle):
    g = load_golden("c2_mini")
    case = synth.make_case("C2", step=0, grid=tuple(g["grid"]), n_samples=700)
    assert _sha(case.traveltimes) == str(g["lut_sha256"])
    assert np.array_equal(case.onsets, g["onsets"])
    a, b, c = oracle.detect(case.onsets, case.traveltimes, case.fsmp, case.lsmp,
                            case.available, threads=4)
    assert np.array_equal(c, g["max_coa_idx"])
    np.testing.assert_allclose(a, g["max_coa"], rtol=1e-14)
    np.testing.assert_allclose(b, g["max_norm_coa"], rtol=1e-14)
    # the injected events are what the scan finds
    for (ijk, t0) in case.event_nodes:
        assert c[t0] == np.ravel_multi_index(ijk, case.grid)


def test_c2_mini_quiet_all_index_zero(oracle):
    g = load_golden("c2_mini_quiet")
    case = synth.make_case("C2", step=1, grid=tuple(g["grid"]), n_samples=200,
                           quiet=True)
    assert _sha(case.traveltimes) == str(g["lut_sha256"])
    a, b, c = oracle.detect(case.onsets, case.traveltimes, case.fsmp, case.lsmp,
                            case.available, threads=4)
    assert (c == 0).all() and np.array_equal(c, g["max_coa_idx"])
    np.testing.assert_allclose(a, g["max_coa"], rtol=1e-14)
    np.testing.assert_allclose(b, g["max_norm_coa"], rtol=1e-14)


def test_c1_icequake_geometry(oracle):
    g = load_golden("c1_icequake_geometry")
    case = synth.make_case("C1", step=0)
    assert _sha(case.traveltimes) == str(g["lut_sha256"])
    assert np.array_equal(case.onsets, g["onsets"])
    a, b, c = oracle.detect(case.onsets, case.traveltimes, case.fsmp, case.lsmp,
                            case.available, threads=8, max_bytes=1 << 29)
    assert np.array_equal(c, g["max_coa_idx"])
    np.testing.assert_allclose(a, g["max_coa"], rtol=1e-14)
    np.testing.assert_allclose(b, g["max_norm_coa"], rtol=1e-14)


def test_stalta_known_answers_and_reference_vectors(oracle):
    g = load_golden("stalta")
    toy = g["toy"]
    # reference tests/test_onsets.py:27-35
    assert (oracle.c_overlapping_sta_lta(toy, 2, 3)
            == np.array([1.0, 1.0, 1.5, 1.25, 21.0 / 18, 27.0 / 24])).all()
    assert np.allclose(oracle.c_centred_sta_lta(toy, 2, 3),
                       np.array([1.0, 1.0, 3.5, 2.25, 1.0, 1.0]))
    for kind in ("overlapping", "centred", "recursive"):
        fn = getattr(oracle, f"c_{kind}_sta_lta")
        np.testing.assert_allclose(fn(toy, 2, 3), g[f"toy_{kind}"], rtol=1e-15)
        np.testing.assert_allclose(fn(g["signal"], int(g["nsta"]), int(g["nlta"])),
                                   g[kind], rtol=1e-13)


def test_binding_level_errors(oracle):
    """ValueError semantics of quakemigrate/core/lib.py:105-110."""
    on = np.ones((3, 50))
    with pytest.raises(ValueError, match="Mismatch"):
        oracle.c_migrate(on, np.zeros((2, 2, 2, 4), dtype=np.int32), 2, 3, 3)


def test_table_serving_restatement_matches_reference_lut_class(oracle):
    """np_serve_traveltimes / np_decimate vs LUT.serve_traveltimes / Grid3D.decimate outputs
    recorded from the reference's own class (oracle/make_golden.py section 9)."""
    g = load_golden("serve_traveltimes")
    index = {k: i for i, k in enumerate(g["keys"])}
    picked = [g["grids"][index[k]] for k, v in zip(g["availability_keys"],
                                                    g["availability_values"]) if v == 1]
    assert np.array_equal(oracle.np_serve_traveltimes(picked, 50), g["served_50"])
    dec = [oracle.np_decimate(p, g["decimate"]) for p in picked]
    assert dec[0].shape == tuple(int(v) for v in g["dec_node_count"])
    assert np.array_equal(oracle.np_serve_traveltimes(dec, 250), g["served_dec_250"])
    # NaN / infinities / beyond-int32 travel times: the recorded cast results of the reference's
    # class (INT32_MIN for what int32 cannot hold; x86-64 semantics, the fixture names its machine)
    nf = load_golden("serve_nonfinite")
    if str(nf["machine"]) == __import__("platform").machine():
        with np.errstate(invalid="ignore"):
            assert np.array_equal(oracle.np_serve_traveltimes(list(nf["grids"]), 50), nf["served_50"])
    assert (nf["served_50"][np.isnan(nf["grids"]).transpose(1, 2, 3, 0)] == np.iinfo(np.int32).min).all()


def _onset_stage_inputs(g, tf):
    """Signals and device-stage transform for the reference's four ``signal_transform`` values:
    the envelope transforms take |hilbert(x)| (computed upstream, like the filters) as the signal
    and then are ``abs`` / ``energy`` of it (stalta.py:518-521)."""
    if tf in ("env", "env_squared"):
        return g["envelopes"], ("abs" if tf == "env" else "energy")
    return g["signals"], tf


def test_onset_stage_restatement_matches_reference_stalta_onset(oracle):
    """np_onset_stage (on the oracle's C STA/LTA port) == the reference's own
    STALTAOnset._onset / _trim_taper_pad (fixture made by running signal/onsets/stalta.py:491-583,
    oracle/make_golden.py section 10), all four transforms and both window positions."""
    from scipy.signal import hilbert

    g = load_golden("onset_stage")
    np.testing.assert_allclose(np.abs(hilbert(g["signals"], axis=-1)), g["envelopes"], rtol=1e-12)
    for pos in ("classic", "centred"):
        for tf in ("energy", "abs", "env", "env_squared"):
            sig, stage_tf = _onset_stage_inputs(g, tf)
            raw, logged = oracle.np_onset_stage(sig, g["trace_row"], g["nsta"], g["nlta"],
                                                stage_tf, pos, int(g["taper_pad"]),
                                                float(g["min_onset_value"]))
            np.testing.assert_allclose(raw, g[f"raw_{pos}_{tf}"], rtol=1e-12)
            np.testing.assert_allclose(logged, g[f"log_{pos}_{tf}"], rtol=1e-12, atol=1e-14)
    raw, _ = oracle.np_onset_stage(g["signals"], g["trace_row"], g["nsta"], g["nlta"], "energy",
                                   "classic", -1, 0.01)
    np.testing.assert_allclose(raw, g["raw_classic_energy_notaper"], rtol=1e-12)


def compute_glue_inputs(g, oracle):
    """Served table, pads and available count of the compute_glue fixture, restated."""
    index = {str(k): i for i, k in enumerate(g["grid_keys"])}
    availability = {str(k): int(v) for k, v in zip(g["availability_keys"],
                                                    g["availability_values"])}
    picked = [g["grids"][index[k]] for k, v in availability.items() if v == 1]
    rate = int(g["sampling_rate"])
    tt = oracle.np_serve_traveltimes(picked, rate)
    fsmp = int(round(float(g["pre_pad"]) * rate))               # util.time2sample
    lsmp = int(round(float(g["post_pad"]) * rate))
    return tt, fsmp, lsmp, availability


def test_oracle_reproduces_reference_compute(oracle):
    """The oracle chain (served table -> migrate -> find_max_coa -> index2grid) == outputs of
    QuakeScan._compute run from the reference's own scan.py:593-647 (fixture compute_glue)."""
    g = load_golden("compute_glue")
    tt, fsmp, lsmp, availability = compute_glue_inputs(g, oracle)
    avail = sum(availability.values())
    a, b, c = oracle.detect(g["onsets"], tt, fsmp, lsmp, avail, threads=2)
    np.testing.assert_allclose(a, g["max_coa"], rtol=1e-14)
    np.testing.assert_allclose(b, g["max_coa_n"], rtol=1e-14)
    ijk = np.column_stack(np.unravel_index(c, tt.shape[:3]))
    assert np.array_equal(g["ll_corner"] + ijk * g["node_spacing"], g["coord"])
    vol = oracle.c_migrate(g["onsets"], tt, fsmp, lsmp, avail, threads=2)
    assert vol.shape == tuple(g["map4d_shape"])
    np.testing.assert_allclose(vol.reshape(-1, vol.shape[-1])[g["map4d_rows"]], g["map4d_vals"],
                               rtol=1e-14)
    assert float(g["detect_time"]) == float(g["starttime"]) + float(g["pre_pad"])


LOCATE_CASES = ["corner", "interior_even", "interior_odd", "near_face", "thin"]


@pytest.mark.parametrize("name", LOCATE_CASES)
def test_locate_fit_restatements_match_reference_calculate_location(oracle, name):
    """np_gaufilt3d / np_gaufit3d / np_covfit3d / np_splineloc vs QuakeScan._calculate_location
    run from the reference's scan.py (fixture locate_fits, make_golden.py section 11)."""
    g = load_golden("locate_fits")
    assert list(g["cases"]) == LOCATE_CASES
    spacing = g[f"{name}_node_spacing"]
    coa = g[f"{name}_map4d"].sum(axis=-1)
    coa = coa / np.nanmax(coa)
    assert np.array_equal(coa, g[f"{name}_coa_map"])
    smoothed = oracle.np_gaufilt3d(coa)
    np.testing.assert_allclose(smoothed, g[f"{name}_smoothed"], rtol=0, atol=1e-14)
    loc, sigma, _ = oracle.np_gaufit3d(smoothed)
    np.testing.assert_allclose(loc, g[f"{name}_gaussian"], rtol=1e-9)
    np.testing.assert_allclose(sigma * spacing, g[f"{name}_gaussian_uncertainty"], rtol=1e-9)
    mean, cov = oracle.np_covfit3d(coa, spacing)
    np.testing.assert_allclose(mean, g[f"{name}_covariance"], rtol=1e-13)
    np.testing.assert_allclose(np.diag(np.sqrt(np.abs(cov))),
                               g[f"{name}_covariance_uncertainty"], rtol=1e-13, atol=1e-300)
    assert np.array_equal(oracle.np_splineloc(coa), g[f"{name}_spline"])


def test_exp_rule_restatement_matches_the_references_scalar_build(oracle):
    """The arg-max rule of the opt-in tie_rule = 1 (oracle.np_argmax_exp_rule: correctly rounded
    exp, first maximum) reproduces the reference's own two C files built with -fno-tree-vectorize
    (glibc's scalar exp) on EVERY sample of the two near-tie families -- and differs from the
    -Ofast build (libmvec) on ~8 % of them, as that build differs from its scalar twin."""
    g = load_golden("permuted_twins")
    t = load_golden("near_ties_scalar")
    idx = oracle.np_argmax_exp_rule(g["onsets"], g["traveltimes"], int(g["fsmp"]), int(g["lsmp"]),
                                    int(g["available"]))
    assert np.array_equal(idx, t["permuted_idx_scalar"])
    assert 0.06 < np.mean(idx != g["max_coa_idx"]) < 0.10
    idx = oracle.np_argmax_exp_rule(t["mirror_onsets"], t["mirror_traveltimes"], int(t["mirror_fsmp"]),
                                    int(t["mirror_lsmp"]), int(t["mirror_available"]))
    assert np.array_equal(idx, t["mirror_idx_scalar"])
    assert 0.06 < np.mean(idx != t["mirror_idx_vec"]) < 0.10
    # round 6: the mirror family on a grid of many bricks, 1600 samples (what the per-brick refinement and its
    # sharded form are pinned on); the reference's values of that run beside the oracle's detect
    b = load_golden("near_ties_bricks")
    args = (b["onsets"], b["traveltimes"], int(b["fsmp"]), int(b["lsmp"]), int(b["available"]))
    idx = oracle.np_argmax_exp_rule(*args)
    assert np.array_equal(idx, b["idx_scalar"])
    assert 0.04 < np.mean(idx != b["idx_vec"]) < 0.10
    got = oracle.detect(*args, threads=4)
    np.testing.assert_allclose(got[0], b["max_coa_scalar"], rtol=1e-13)
    np.testing.assert_allclose(got[1], b["max_norm_coa_scalar"], rtol=1e-12)
