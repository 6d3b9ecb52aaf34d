# -*- coding: utf-8 -*-
"""
Locate and load the compiled HIP engine (mirror of the reference's loader,
quakemigrate/core/libnames.py:18-47: same function name, same ImportError
behaviour).  The library lives in-tree at ``quakemigrate_amd/csrc/`` and is
built by ``__graft_entry__.build()``; there is no CPU fallback -- if it cannot
be loaded the import fails loudly.
"""

import ctypes
import pathlib

_CSRC = pathlib.Path(__file__).resolve().parent.parent / "csrc"
LIBRARY_FILE = {"qmlib": "libqmhip.so"}


def _load_cdll(name):
    """
    Load the engine's shared library.

    Parameters
    ----------
    name : str
        Name of library to load (``"qmlib"``, as in the reference).

    Returns
    -------
    cdll : `ctypes.CDLL`
    """
    lib = _CSRC / LIBRARY_FILE.get(name, name)
    try:
        cdll = ctypes.CDLL(str(lib))
    except Exception as e:
        raise ImportError(
            f"Could not load the HIP engine library '{lib}'.\n\n{e}\n\nBuild it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` from the repository "
            "root (needs hipcc, --offload-arch=gfx950). There is no CPU fallback."
        )
    return cdll
