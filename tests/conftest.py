# -*- coding: utf-8 -*-
"""Shared pytest plumbing: markers, paths, golden-vector loading."""

import os
import pathlib
import sys

# The oracle's OpenMP loops run with one thread per logical CPU.  On a box whose CPU share is
# smaller than its CPU count (a container quota, a busy neighbour) libgomp's default of spinning
# at barriers turns that into minutes per call; blocked waits cost nothing measurable here.  Must
# be in the environment before libgomp initialises, i.e. before numpy / torch / the oracle load.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np  # noqa: E402
import pytest  # noqa: E402

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / "tests" / "golden"

# float tolerance of the north-star (BASELINE.json): 1e-6 relative on the
# coalescence values; argmax indices must be identical.
RTOL = 1.0e-6


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


def load_golden(name):
    with np.load(GOLDEN / f"{name}.npz", allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (C restatement); built on demand."""
    import subprocess

    so = ROOT / "oracle" / "libqm_oracle.so"
    if not so.exists():
        subprocess.check_call(["make", "-C", str(ROOT / "oracle"), "libqm_oracle.so"])
    from oracle import qm_oracle

    return qm_oracle
