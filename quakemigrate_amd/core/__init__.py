# -*- coding: utf-8 -*-
"""
``quakemigrate_amd.core`` -- same public names as ``quakemigrate.core``
(quakemigrate/core/__init__.py:21-27), backed by the MI355X HIP engine.
"""

from .lib import (  # noqa: F401
    Engine,
    QMHipError,
    centred_sta_lta,
    default_engine,
    find_max_coa,
    migrate,
    migrate_and_find_max,
    overlapping_sta_lta,
    recursive_sta_lta,
)
