# -*- coding: utf-8 -*-
"""
Locate and load the compiled HIP engine (mirror of the reference's loader,
quakemigrate/core/libnames.py:18-47: same function name, same ImportError
behaviour).  The library lives in-tree at ``quakemigrate_amd/csrc/`` and is
built by ``__graft_entry__.build()``; there is no CPU fallback -- if it cannot
be loaded the import fails loudly.
"""

import ctypes
import importlib.util
import os
import pathlib

_CSRC = pathlib.Path(__file__).resolve().parent.parent / "csrc"
LIBRARY_FILE = {"qmlib": "libqmhip.so"}


def _share_torch_hip_runtime():
    """
    PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` / ``libhsa-runtime64.so``.
    Two HIP runtimes in one process do not share the GPU ("No HIP GPUs are
    available" in whichever initialises second), so when torch is installed its
    copy is loaded first and globally: our library's ``libamdhip64.so.7``
    dependency then binds to that same runtime, and device pointers / streams of
    torch tensors are valid inside the engine.  Without torch the system ROCm
    runtime is used.  (No ``import torch`` here: that costs seconds.)
    """
    if os.environ.get("QM_HIP_SYSTEM_RUNTIME"):
        return None
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    libdir = pathlib.Path(list(spec.submodule_search_locations)[0]) / "lib"
    loaded = None
    for soname in ("libhsa-runtime64.so", "libamdhip64.so"):
        cand = libdir / soname
        if cand.exists():
            try:
                loaded = ctypes.CDLL(str(cand), mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return None
    return loaded


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
    if os.environ.get("QM_HIP_LIB"):            # development: alternative build of the library
        lib = pathlib.Path(os.environ["QM_HIP_LIB"])
    _share_torch_hip_runtime()
    try:
        cdll = ctypes.CDLL(str(lib))
    except Exception as e:
        raise ImportError(
            f"Could not load the HIP engine library '{lib}'.\n\n{e}\n\nBuild it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` from the repository "
            "root (needs hipcc, --offload-arch=gfx950). There is no CPU fallback."
        )
    return cdll
