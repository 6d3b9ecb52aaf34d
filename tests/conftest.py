# -*- coding: utf-8 -*-
"""Shared pytest plumbing: markers, paths, golden-vector loading."""

import pathlib
import sys

import numpy as np
import pytest

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
