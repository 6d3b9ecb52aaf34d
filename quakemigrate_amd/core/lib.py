# -*- coding: utf-8 -*-
"""
Bindings for the MI355X coalescence-migration engine -- the host-side mirror of
the reference's ``quakemigrate/core/lib.py`` for the migrate / find_max_coa path.

Same function names, argument meaning, return values and error behaviour as the
reference (``lib.py:52-125`` ``migrate``, ``lib.py:131-170`` ``find_max_coa``,
``lib.py:176-285`` the three STA/LTA functions), so ``QuakeScan._compute``
(``quakemigrate/signal/scan.py:635-638``) can call them unchanged.  On top of that
``migrate_and_find_max`` is the fused call that never materialises the 4-D map
(what ``detect()`` wants, ``scan.py:641-642``).

Everything numeric below the clip/log pre-processing runs in the HIP library
(``include/qmhip.h``); NumPy is used only for the reference's own host-side
pre-processing (``np.clip`` + ``np.log``, ``lib.py:93-94``) and for allocating
the caller-owned result arrays.
"""

from __future__ import annotations

import ctypes
import logging
import weakref

import numpy as np
import numpy.ctypeslib as clib

from quakemigrate_amd.core.libnames import _load_cdll

qmlib = _load_cdll("qmlib")

c_int32 = ctypes.c_int32
c_int64 = ctypes.c_int64
c_dPt = clib.ndpointer(dtype=np.double, flags="C_CONTIGUOUS")
c_i32Pt = clib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
c_i64Pt = clib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")

stalta_header_t = np.dtype(
    [("n", c_int32), ("nsta", c_int32), ("nlta", c_int32)], align=True
)
stalta_header_pt = clib.ndpointer(stalta_header_t, flags="C_CONTIGUOUS")

# ---- reference-compatible symbols (qmlib.h:28-44) ---------------------------
qmlib.migrate.argtypes = [c_dPt, c_i32Pt, c_dPt, c_int32, c_int32, c_int32,
                          c_int32, c_int32, c_int64, c_int64]
qmlib.migrate.restype = None
qmlib.find_max_coa.argtypes = [c_dPt, c_dPt, c_dPt, c_i64Pt, c_int32, c_int64,
                               c_int64]
qmlib.find_max_coa.restype = None
for _f in (qmlib.overlapping_sta_lta, qmlib.centred_sta_lta,
           qmlib.recursive_sta_lta):
    _f.argtypes = [c_dPt, stalta_header_pt, c_dPt]
    _f.restype = None

# ---- handle API (include/qmhip.h part 2) ------------------------------------
_vp = ctypes.c_void_p
qmlib.qm_last_error.restype = ctypes.c_char_p
qmlib.qm_device_count.restype = ctypes.c_int
qmlib.qm_build_info.restype = ctypes.c_char_p
qmlib.qm_build_info.argtypes = []
qmlib.qm_compat_status.restype = ctypes.c_int
qmlib.qm_table_hash.restype = None
qmlib.qm_table_hash.argtypes = [ctypes.c_void_p, c_int64, ctypes.POINTER(ctypes.c_uint64),
                                ctypes.POINTER(ctypes.c_uint64)]
qmlib.qm_engine_create.argtypes = [ctypes.c_int, ctypes.POINTER(_vp)]
qmlib.qm_release_cached_memory.argtypes = []
qmlib.qm_release_cached_memory.restype = ctypes.c_int
qmlib.qm_engine_destroy.argtypes = [_vp]
qmlib.qm_engine_destroy.restype = None
qmlib.qm_engine_set_stream.argtypes = [_vp, _vp, ctypes.c_int]
qmlib.qm_engine_synchronize.argtypes = [_vp]
qmlib.qm_engine_config.argtypes = [_vp, ctypes.c_char_p, c_int64]
qmlib.qm_engine_get.argtypes = [_vp, ctypes.c_char_p, ctypes.POINTER(c_int64)]
qmlib.qm_engine_load_lut.argtypes = [_vp, _vp, ctypes.c_int, c_int32, c_int32,
                                     c_int32, c_int32, c_int64]
qmlib.qm_engine_table_select.argtypes = [_vp, ctypes.c_uint64, c_int32, ctypes.POINTER(c_int32)]
qmlib.qm_engine_lut_max.argtypes = [_vp, ctypes.POINTER(c_int32)]
qmlib.qm_engine_grids_begin.argtypes = [_vp, c_int32, c_int32, c_int32, c_int32]
qmlib.qm_engine_grids_set.argtypes = [_vp, c_int32, _vp, ctypes.c_int]
qmlib.qm_engine_serve.argtypes = [_vp, ctypes.c_double, c_i32Pt, c_int32, c_int32, c_int32,
                                  c_int32, c_int64]
qmlib.qm_engine_lut_download.argtypes = [_vp, c_i32Pt]
qmlib.qm_engine_detect.argtypes = [_vp, _vp, ctypes.c_int, c_int32, c_int32,
                                   c_int32, c_int32, c_int64, _vp, _vp, _vp,
                                   ctypes.c_int]
qmlib.qm_engine_detect_batch.argtypes = [_vp, _vp, ctypes.c_int, c_int32, c_int32, c_int32,
                                         c_int32, c_int32, c_int64, _vp, _vp, _vp,
                                         ctypes.c_int]
qmlib.qm_engine_detect_partial.argtypes = [_vp, _vp, ctypes.c_int, c_int32,
                                           c_int32, c_int32, c_int32, _vp, _vp,
                                           _vp]
qmlib.qm_engine_finalize.argtypes = [_vp, _vp, _vp, _vp, c_int32, c_int32,
                                     c_int64, _vp, _vp, _vp, ctypes.c_int]
qmlib.qm_engine_finalize_packed.argtypes = [_vp, _vp, c_int32, c_int32, c_int64, _vp, _vp,
                                            _vp, ctypes.c_int]
qmlib.qm_engine_tie_partial.argtypes = [_vp, _vp, ctypes.c_int, c_int32, c_int32, c_int32, c_int32, _vp,
                                        c_int32, _vp]
qmlib.qm_engine_tie_fold.argtypes = [_vp, _vp, c_int32, c_int32, _vp]
qmlib.qm_engine_migrate.argtypes = [_vp, _vp, ctypes.c_int, c_int32, c_int32,
                                    c_int32, c_int32, c_int64, _vp, ctypes.c_int,
                                    ctypes.c_int, _vp, _vp, _vp, ctypes.c_int]
qmlib.qm_engine_marginal.argtypes = [_vp, _vp, ctypes.c_int, c_int32, c_int32, c_int32,
                                     c_int32, c_int64, c_int32, c_int32, _vp, ctypes.c_int,
                                     _vp, _vp, _vp, ctypes.c_int]
qmlib.qm_engine_locate_fits.argtypes = [_vp, _vp, ctypes.c_int, c_int32, c_int32, c_int32,
                                        ctypes.c_double, ctypes.c_double,
                                        ctypes.POINTER(ctypes.c_double), _vp, _vp, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double)]
qmlib.qm_engine_rbf_peak.argtypes = [_vp, ctypes.POINTER(ctypes.c_double), c_int32, c_int32,
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int64)]
qmlib.qm_engine_onsets.argtypes = [_vp, _vp, ctypes.c_int, c_int32, c_int32, c_i32Pt, c_int32,
                                   c_i32Pt, c_i32Pt, ctypes.c_int, ctypes.c_int, c_int32,
                                   ctypes.c_double, _vp, _vp, ctypes.c_int]
qmlib.qm_engine_find_max_coa.argtypes = [_vp, _vp, ctypes.c_int, c_int32,
                                         c_int64, _vp, _vp, _vp, ctypes.c_int]
qmlib.qm_exp2f_max_error.argtypes = [_vp, ctypes.c_float, ctypes.c_float,
                                     ctypes.POINTER(ctypes.c_double)]
qmlib.qm_exp_correctly_rounded.argtypes = [ctypes.c_double]
qmlib.qm_exp_correctly_rounded.restype = ctypes.c_double
qmlib.qm_engine_exp_correctly_rounded.argtypes = [_vp, c_dPt, c_int64, c_dPt]
qmlib.qm_stream_create.argtypes = [_vp, c_int32, c_int32, c_int32, c_int32, c_int64, c_int32, c_int32,
                                   ctypes.POINTER(_vp)]
qmlib.qm_stream_destroy.argtypes = [_vp]
qmlib.qm_stream_destroy.restype = None
qmlib.qm_stream_push.argtypes = [_vp, _vp]
qmlib.qm_stream_flush.argtypes = [_vp]
qmlib.qm_stream_pop.argtypes = [_vp, c_int32, _vp, _vp, _vp]
qmlib.qm_stream_pending.argtypes = [_vp, ctypes.POINTER(c_int32), ctypes.POINTER(c_int32)]
qmlib.qm_engine_last_kernel_ms.argtypes = [_vp, ctypes.POINTER(ctypes.c_double)]
qmlib.qm_engine_kernel_log.argtypes = [_vp, ctypes.POINTER(ctypes.c_double),
                                       ctypes.POINTER(c_int32)]


class QMHipError(RuntimeError):
    """A call into the HIP engine failed (message from ``qm_last_error``)."""


def _check(rc):
    if rc != 0:
        raise QMHipError(qmlib.qm_last_error().decode(errors="replace"))


def _host(a):
    return a.ctypes.data_as(_vp)


def build_info():
    """What the loaded library was built from (include/qmhip.h: qm_build_info): the generated shift-reuse
    loops' constants and generator digest, overlay and development defines ("none" in the product build)."""
    return qmlib.qm_build_info().decode()


class Engine:
    """
    One GPU's migration engine (thin object wrapper over the C handle API).

    Pointers may be NumPy arrays (host, synchronous) or integers / objects with
    ``data_ptr()`` (device memory on this engine's GPU, asynchronous on the
    engine's stream) -- see :func:`_ptr`.
    """

    def __init__(self, device=0, **config):
        h = _vp()
        _check(qmlib.qm_engine_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)
        self._finalizer = weakref.finalize(self, qmlib.qm_engine_destroy, h)
        self.grid = None
        self.n_rows = None
        self.node_offset = 0
        self.table_generation = 0       # bumped whenever the resident table is replaced
        self._current_key = None        # key of the resident table (select_table), if it has one
        self.grids_generation = 0       # bumped by set_traveltime_grids (the grids are shared by
                                        # every MigrationScan on this engine)
        for k, v in config.items():
            self.config(k, v)

    # -- plumbing -----------------------------------------------------------
    def close(self):
        self._finalizer()

    def config(self, key, value):
        _check(qmlib.qm_engine_config(self._h, key.encode(), int(value)))

    def get(self, key):
        v = c_int64()
        _check(qmlib.qm_engine_get(self._h, key.encode(), ctypes.byref(v)))
        return int(v.value)

    def set_stream(self, stream_ptr):
        """
        Run on the given ``hipStream_t`` handle (an int; 0 is the device's default stream,
        i.e. ``torch.cuda.current_stream().cuda_stream`` outside a stream context), or on
        the engine's private stream with ``None``.
        """
        if stream_ptr is None:
            _check(qmlib.qm_engine_set_stream(self._h, _vp(None), 1))
        else:
            _check(qmlib.qm_engine_set_stream(self._h, _vp(int(stream_ptr) or None), 0))

    def synchronize(self):
        _check(qmlib.qm_engine_synchronize(self._h))

    def last_kernel_ms(self):
        ms = ctypes.c_double()
        _check(qmlib.qm_engine_last_kernel_ms(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def exp2f_max_error(self, lo, hi):
        """Largest relative deviation of the GPU's ``v_exp_f32`` from the float64 ``exp2`` over
        every float32 in ``[lo, hi]`` (self-check behind the screened detect's error bound)."""
        out = ctypes.c_double()
        _check(qmlib.qm_exp2f_max_error(self._h, float(lo), float(hi), ctypes.byref(out)))
        return float(out.value)

    def exp_correctly_rounded(self, x):
        """``exp(x)`` rounded to nearest, evaluated on the GPU (csrc/qm_ties.hpp: what the opt-in
        ``tie_rule = 1`` compares near-tied nodes on)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        _check(qmlib.qm_engine_exp_correctly_rounded(self._h, x.reshape(-1), x.size, out.reshape(-1)))
        return out

    def kernel_log(self):
        """(total ms, launches) of the stacking kernels since the last call."""
        ms, n = ctypes.c_double(), c_int32()
        _check(qmlib.qm_engine_kernel_log(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)

    _TORCH_NAMES = {np.dtype(np.float64): "torch.float64", np.dtype(np.int64): "torch.int64",
                    np.dtype(np.int32): "torch.int32"}

    def _ptr(self, x, dtype=None, count=None):
        """
        (void*, on_device) for a NumPy array, a torch tensor or a raw int.  Arrays and tensors
        are checked: element type ``dtype``, contiguity, at least ``count`` elements, and -- for
        a device tensor -- that it lives on this engine's GPU (a wrong type or device would be
        reinterpreted silently by the C side).  A raw int is taken as a device address as is.
        """
        if isinstance(x, np.ndarray):
            if dtype is not None and x.dtype != dtype:
                raise TypeError(f"expected {np.dtype(dtype)}, got {x.dtype}")
            if not x.flags["C_CONTIGUOUS"]:
                raise ValueError("array must be C-contiguous")
            if count is not None and x.size < count:
                raise ValueError(f"array holds {x.size} elements, {count} needed")
            return _host(x), 0
        if hasattr(x, "data_ptr"):                  # torch tensor
            if not x.is_contiguous():
                raise ValueError("tensor must be contiguous")
            if dtype is not None and str(x.dtype) != self._TORCH_NAMES[np.dtype(dtype)]:
                raise TypeError(f"expected {self._TORCH_NAMES[np.dtype(dtype)]}, got {x.dtype}")
            if count is not None and x.numel() < count:
                raise ValueError(f"tensor holds {x.numel()} elements, {count} needed")
            if x.is_cuda and x.device.index != self.device:
                raise ValueError(f"tensor lives on cuda:{x.device.index}, this engine runs on "
                                 f"cuda:{self.device}")
            return _vp(x.data_ptr()), (1 if x.is_cuda else 0)
        return _vp(int(x)), 1

    def _series(self, out, n):
        """Pointers of a (float64, float64, int64) output triple of >= n elements each; the
        three must be all host or all device."""
        (pa, da), (pb, db), (pc, dc) = (self._ptr(out[0], np.float64, n),
                                        self._ptr(out[1], np.float64, n),
                                        self._ptr(out[2], np.int64, n))
        if not da == db == dc:
            raise ValueError("the three output series must be all host or all device")
        return pa, pb, pc, da

    # -- table --------------------------------------------------------------
    def load_lut(self, traveltimes, node_offset=0, shape=None):
        """
        Make the int32 travel-time table resident: shape (nx, ny, nz, n_rows), C
        order (what ``LUT.serve_traveltimes`` returns, lut.py:538).
        """
        if shape is None:
            shape = tuple(traveltimes.shape)
        if len(shape) != 4:
            raise ValueError("traveltimes must have shape (nx, ny, nz, n_rows)")
        p, dev = self._ptr(traveltimes, np.int32)
        nx, ny, nz, rows = (int(v) for v in shape)
        self._foreign_load()
        _check(qmlib.qm_engine_load_lut(self._h, p, dev, nx, ny, nz, rows,
                                        int(node_offset)))
        self.grid = (nx, ny, nz)
        self.n_rows = rows
        self.node_offset = int(node_offset)
        self.table_generation += 1

    def _foreign_load(self):
        """A load on top of a resident table (no ``select_table`` miss in between) replaces it and
        does not inherit its key -- the C side un-keys it the same way (qm_engine_load_lut)."""
        if self.grid is not None:
            self._current_key = None

    def select_table(self, key, capacity=4):
        """
        Work on the table known under ``key`` (any hashable): parks the current table's device
        state -- up to ``capacity`` tables, least recently used evicted -- and brings the one parked
        under ``key`` back (a pointer swap).  Returns True if that table is resident now; False if
        the caller has to load it (``load_lut`` / ``serve``: it is then known under ``key``).
        For the alternating station availabilities of a continuous run (lut.py:529-537).
        """
        import hashlib

        digest = hashlib.blake2b(repr(key).encode(), digest_size=8).digest()
        k64 = int.from_bytes(digest, "little")
        if not hasattr(self, "_tables"):
            self._tables = {}
        if self._current_key is not None:
            self._tables.pop(self._current_key, None)           # (most recent last)
            self._tables[self._current_key] = (self.grid, self.n_rows, self.node_offset)
            while len(self._tables) > 256:                      # shapes of long-evicted tables
                self._tables.pop(next(iter(self._tables)))
        resident = c_int32()
        _check(qmlib.qm_engine_table_select(self._h, k64, int(capacity), ctypes.byref(resident)))
        self._current_key = k64
        if resident.value and k64 in self._tables:
            self.grid, self.n_rows, self.node_offset = self._tables[k64]
        elif not resident.value:
            self.grid, self.n_rows = None, None
        self.table_generation += 1
        return bool(resident.value)

    # -- on-device serving (lut.py:502-538 / :102-140 moved to the GPU) ----------
    def set_traveltime_grids(self, grids):
        """
        Upload float64 travel-time grids in seconds, each (nx, ny, nz): one per
        station/phase, e.g. ``[lut[station][phase] for ...]``.  Done once per LUT.
        """
        grids = list(grids)
        nx, ny, nz = (int(v) for v in grids[0].shape)
        self.grids_generation += 1
        _check(qmlib.qm_engine_grids_begin(self._h, nx, ny, nz, len(grids)))
        for i, g in enumerate(grids):
            if tuple(g.shape) != (nx, ny, nz):
                raise ValueError("all travel-time grids must share one shape")
            if isinstance(g, np.ndarray):
                g = np.ascontiguousarray(g, dtype=np.float64)
            p, dev = self._ptr(g, np.float64)
            _check(qmlib.qm_engine_grids_set(self._h, i, p, dev))

    def serve(self, sampling_rate, rows, decimate=(1, 1, 1), node_offset=0):
        """
        Build and make resident the int32 table ``rint(grid[rows[s]] * sampling_rate)`` of
        the selected grids, decimated like ``Grid3D.decimate``.
        """
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        dfx, dfy, dfz = (int(v) for v in decimate)
        self._foreign_load()
        _check(qmlib.qm_engine_serve(self._h, float(sampling_rate), rows, len(rows), dfx, dfy,
                                     dfz, int(node_offset)))
        self.grid = (self.get("nx"), self.get("ny"), self.get("nz"))
        self.table_generation += 1
        self.n_rows = len(rows)
        self.node_offset = int(node_offset)

    def download_lut(self):
        out = np.empty(tuple(self.grid) + (self.n_rows,), dtype=np.int32)
        _check(qmlib.qm_engine_lut_download(self._h, out))
        return out

    @property
    def lut_max(self):
        v = c_int32()
        _check(qmlib.qm_engine_lut_max(self._h, ctypes.byref(v)))
        return int(v.value)

    @property
    def n_nodes(self):
        return int(np.prod(self.grid))

    # -- steps --------------------------------------------------------------
    def detect(self, log_onsets, fsmp, lsmp, available, n_nodes_total=None,
               out=None):
        """Fused migrate + find_max_coa; ``log_onsets`` already log(clip(.))."""
        rows, t_samples = (int(v) for v in log_onsets.shape)
        self._check_rows(rows)
        n = t_samples - fsmp - lsmp
        if out is None:
            out = (np.zeros(max(n, 0)), np.zeros(max(n, 0)),
                   np.zeros(max(n, 0), dtype=np.int64))
        po, dev_on = self._ptr(log_onsets, np.float64)
        pa, pb, pc, da = self._series(out, max(n, 0))
        total = self.n_nodes if n_nodes_total is None else int(n_nodes_total)
        _check(qmlib.qm_engine_detect(self._h, po, dev_on, t_samples, int(fsmp),
                                      int(lsmp), int(available), total, pa, pb,
                                      pc, da))
        return out

    def detect_batch(self, log_onsets, fsmp, lsmp, available, n_nodes_total=None, out=None):
        """
        ``K`` consecutive timesteps of the detect sweep in one launch: ``log_onsets`` (K, rows, T),
        already ``log(clip(.))``; returns / fills three (K, n_samples) series.  Every step's result
        is what :meth:`detect` gives for it (``get("steps_per_launch")`` tells whether the kernels
        took the K steps in one launch or the call went step by step).
        """
        steps, rows, t_samples = (int(v) for v in log_onsets.shape)
        self._check_rows(rows)
        n = max(t_samples - int(fsmp) - int(lsmp), 0)
        if out is None:
            out = (np.zeros((steps, n)), np.zeros((steps, n)), np.zeros((steps, n), dtype=np.int64))
        po, dev_on = self._ptr(log_onsets, np.float64)
        pa, pb, pc, da = self._series(out, steps * n)
        total = self.n_nodes if n_nodes_total is None else int(n_nodes_total)
        _check(qmlib.qm_engine_detect_batch(self._h, po, dev_on, steps, t_samples, int(fsmp),
                                            int(lsmp), int(available), total, pa, pb, pc, da))
        return out

    def detect_partial(self, log_onsets, fsmp, lsmp, available, part):
        """``part`` = (max, idx, sum) device buffers of length n_samples."""
        rows, t_samples = (int(v) for v in log_onsets.shape)
        self._check_rows(rows)
        po, dev_on = self._ptr(log_onsets, np.float64)
        n = max(t_samples - int(fsmp) - int(lsmp), 0)
        (pm, dm), (pi, di), (ps, ds) = (self._ptr(part[0], np.float64, n),
                                        self._ptr(part[1], np.int64, n),
                                        self._ptr(part[2], np.float64, n))
        if not (dm and di and ds):
            raise ValueError("detect_partial writes device buffers: pass tensors on this GPU")
        _check(qmlib.qm_engine_detect_partial(
            self._h, po, dev_on, t_samples, int(fsmp), int(lsmp), int(available), pm, pi, ps))

    def finalize(self, part_max, part_idx, part_sum, n_sets, n_samples,
                 n_nodes_total, out=None):
        if out is None:
            out = (np.zeros(n_samples), np.zeros(n_samples),
                   np.zeros(n_samples, dtype=np.int64))
        pa, pb, pc, da = self._series(out, int(n_samples))
        need = int(n_sets) * int(n_samples)
        _check(qmlib.qm_engine_finalize(
            self._h, self._ptr(part_max, np.float64, need)[0],
            self._ptr(part_idx, np.int64, need)[0], self._ptr(part_sum, np.float64, need)[0],
            int(n_sets), int(n_samples), int(n_nodes_total), pa, pb, pc, da))
        return out

    def finalize_packed(self, packed, n_sets, n_samples, n_nodes_total, out=None):
        """``packed``: device float64 [n_sets][3][n_samples] (maxima, int64 index bits, sums) --
        the all-gathered partials of a sharded detect; returns the final series."""
        if out is None:
            out = (np.zeros(n_samples), np.zeros(n_samples),
                   np.zeros(n_samples, dtype=np.int64))
        pa, pb, pc, da = self._series(out, int(n_samples))
        pp, dev = self._ptr(packed, np.float64, 3 * int(n_sets) * int(n_samples))
        if not dev:
            raise ValueError("finalize_packed reads a device buffer")
        _check(qmlib.qm_engine_finalize_packed(self._h, pp, int(n_sets), int(n_samples),
                                               int(n_nodes_total), pa, pb, pc, da))
        return out

    def tie_partial(self, log_onsets, fsmp, lsmp, available, packed, n_sets, tie_packed):
        """``tie_rule = 1`` on a sharded detect: this engine's near-tie candidates of the step its last
        :meth:`detect_partial` computed, against the grid's maxima in ``packed`` (the gathered partials,
        device float64 [n_sets][3][n_samples]) -> ``tie_packed`` device float64 [2][n_samples] (bit
        patterns: largest correctly rounded exp, lowest global index reaching it)."""
        rows, t_samples = (int(v) for v in log_onsets.shape)
        self._check_rows(rows)
        po, dev_on = self._ptr(log_onsets, np.float64)
        n = max(t_samples - int(fsmp) - int(lsmp), 0)
        pp, dev = self._ptr(packed, np.float64, 3 * int(n_sets) * n)
        pt, dev_t = self._ptr(tie_packed, np.float64, 2 * n)
        if not (dev and dev_t):
            raise ValueError("tie_partial reads and writes device buffers")
        _check(qmlib.qm_engine_tie_partial(self._h, po, dev_on, t_samples, int(fsmp), int(lsmp),
                                           int(available), pp, int(n_sets), pt))

    def tie_fold(self, tie_gathered, n_sets, n_samples, idx):
        """The gathered ``[n_sets][2][n_samples]`` outcomes of :meth:`tie_partial` folded into the
        device index series ``idx`` (in place)."""
        pg, dev = self._ptr(tie_gathered, np.float64, 2 * int(n_sets) * int(n_samples))
        pi, dev_i = self._ptr(idx, np.int64, int(n_samples))
        if not (dev and dev_i):
            raise ValueError("tie_fold reads and writes device buffers")
        _check(qmlib.qm_engine_tie_fold(self._h, pg, int(n_sets), int(n_samples), pi))

    def migrate(self, log_onsets, fsmp, lsmp, available, map4d, scan_out=None,
                accumulate=False, n_nodes_total=None):
        rows, t_samples = (int(v) for v in log_onsets.shape)
        self._check_rows(rows)
        po, dev_on = self._ptr(log_onsets, np.float64)
        n = max(t_samples - int(fsmp) - int(lsmp), 0)
        pm, dev_map = self._ptr(map4d, np.float64, self.n_nodes * n)
        if scan_out is None:
            pa = pb = pc = _vp(None)
            da = 0
        else:
            pa, pb, pc, da = self._series(scan_out, n)
        total = self.n_nodes if n_nodes_total is None else int(n_nodes_total)
        _check(qmlib.qm_engine_migrate(
            self._h, po, dev_on, t_samples, int(fsmp), int(lsmp), int(available),
            total, pm, dev_map, 1 if accumulate else 0, pa, pb, pc, da))
        return map4d

    def marginal_map(self, log_onsets, fsmp, lsmp, available, first_sample, end_sample,
                     out=None, scan_out=None, n_nodes_total=None):
        """
        Sum over scanned samples ``[first_sample, end_sample)`` of every node's coalescence,
        shape ``grid`` -- ``np.sum(map4d[..., first:end], axis=-1)`` without the 4-D map.
        """
        rows, t_samples = (int(v) for v in log_onsets.shape)
        self._check_rows(rows)
        if out is None:
            out = np.zeros(self.grid, dtype=np.float64)
        po, dev_on = self._ptr(log_onsets, np.float64)
        pm, dev_map = self._ptr(out, np.float64, self.n_nodes)
        if scan_out is None:
            pa = pb = pc = _vp(None)
            da = 0
        else:
            pa, pb, pc, da = self._series(scan_out, max(t_samples - int(fsmp) - int(lsmp), 0))
        total = self.n_nodes if n_nodes_total is None else int(n_nodes_total)
        _check(qmlib.qm_engine_marginal(
            self._h, po, dev_on, t_samples, int(fsmp), int(lsmp), int(available), total,
            int(first_sample), int(end_sample), pm, dev_map, pa, pb, pc, da))
        return out

    def locate_fits(self, coa_map, node_spacing, sgm=0.8, cov_thresh=0.90, norm_out=None,
                    smoothed_out=None):
        """
        Device part of ``QuakeScan._calculate_location`` (scan.py:696-733) on a marginalised
        map ``coa_map`` (nx, ny, nz), host array or device tensor: normalisation, the two-pass
        Gaussian smoothing, the covariance moments and the two fit windows.  Returns a dict;
        ``norm_out`` / ``smoothed_out`` (both host or both device) receive the maps.
        """
        nx, ny, nz = (int(v) for v in coa_map.shape)
        pm, dev_map = self._ptr(coa_map, np.float64)
        outs = [o for o in (norm_out, smoothed_out) if o is not None]
        pn = ps = _vp(None)
        dev_out = 0
        if norm_out is not None:
            pn, dev_out = self._ptr(norm_out, np.float64)
        if smoothed_out is not None:
            ps, dev_s = self._ptr(smoothed_out, np.float64)
            if len(outs) == 2 and dev_s != dev_out:
                raise ValueError("norm_out and smoothed_out must both be host or both device")
            dev_out = dev_s
        spacing = (ctypes.c_double * 3)(*[float(v) for v in node_spacing])
        summary = (ctypes.c_double * 16)()
        gau = np.zeros((7, 7, 7))
        spl = np.zeros((5, 5, 5))
        dp = ctypes.POINTER(ctypes.c_double)
        _check(qmlib.qm_engine_locate_fits(
            self._h, pm, dev_map, nx, ny, nz, float(sgm), float(cov_thresh), spacing, pn, ps,
            dev_out, summary, gau.ctypes.data_as(dp), spl.ctypes.data_as(dp)))
        sm = np.array(summary[:])
        cov = np.array([[sm[8], sm[11], sm[12]], [sm[11], sm[9], sm[13]],
                        [sm[12], sm[13], sm[10]]])
        return {"map_max": sm[0],
                "peak": np.array(np.unravel_index(int(sm[1]), (nx, ny, nz))),
                "smoothed_mean": sm[2],
                "smoothed_peak": np.array(np.unravel_index(int(sm[3]), (nx, ny, nz))),
                "weight": sm[4], "expectation": sm[5:8].copy(), "covariance": cov,
                "pass_maxima": sm[14:16].copy(), "gaussian_window": gau,
                "spline_window": spl}

    def rbf_peak(self, weights, upscale=10):
        """
        First maximum of the cubic radial-basis interpolant with the given ``weights`` (n, n, n)
        on the ``upscale``-times finer grid (``_splineloc``'s refinement, scan.py:777-812),
        evaluated on the GPU.  Returns ``(value, (i, j, k))`` -- indices into the fine grid of
        ``(n - 1) * upscale + 1`` points per axis.
        """
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if w.ndim != 3 or not (w.shape[0] == w.shape[1] == w.shape[2]):
            raise ValueError("weights must be a cube")
        n = int(w.shape[0])
        value, index = ctypes.c_double(), c_int64()
        _check(qmlib.qm_engine_rbf_peak(self._h, w.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                        n, int(upscale), ctypes.byref(value), ctypes.byref(index)))
        m = (n - 1) * int(upscale) + 1
        return float(value.value), tuple(int(v) for v in np.unravel_index(int(index.value), (m, m, m)))

    def onsets(self, signals, trace_row, nsta, nlta, transform="energy", position="classic",
               taper_pad=-1, min_onset_value=0.4, raw_out=None, log_out=None):
        """
        Onset stage on the GPU (``STALTAOnset._onset`` + the clip/log of ``lib.migrate``): from
        pre-processed component waveforms ``signals`` (n_traces, T) to the raw onset rows and
        ``log(clip(onset, 0.01))`` rows, shape (n_rows, T).  Returns ``(raw, logged)``.

        ``transform``: the reference's four ``signal_transform`` values (stalta.py:515-521).
        ``"env"`` / ``"env_squared"`` take the envelope ``|hilbert(x)|`` first -- an upstream
        transform like the filters, done here with the reference's own ``scipy.signal.hilbert``
        call on host signals (pass the envelope yourself with ``"abs"`` / ``"energy"`` for
        device-resident signals) -- the STA/LTA and everything after it run on the GPU.
        ``position``: ``"classic"``, ``"centred"`` or ``"recursive"`` (onsetlib.c:126-148).
        """
        if transform in ("env", "env_squared"):
            if not isinstance(signals, np.ndarray):
                raise ValueError("env transforms of device-resident signals: pass the envelope "
                                 "|hilbert(x)| with transform='abs' / 'energy'")
            from scipy.signal import hilbert

            signals = np.ascontiguousarray(np.abs(hilbert(signals, axis=-1)))
            transform = "abs" if transform == "env" else "energy"
        n_traces, t_samples = (int(v) for v in signals.shape)
        trace_row = np.ascontiguousarray(trace_row, dtype=np.int32)
        nsta = np.ascontiguousarray(nsta, dtype=np.int32)
        nlta = np.ascontiguousarray(nlta, dtype=np.int32)
        n_rows = len(nsta)
        ps, dev_s = self._ptr(signals, np.float64)
        if log_out is None:
            log_out = np.zeros((n_rows, t_samples))
        if raw_out is None and isinstance(log_out, np.ndarray):
            raw_out = np.zeros((n_rows, t_samples))
        pl, dev_o = self._ptr(log_out, np.float64, n_rows * t_samples)
        pr = (self._ptr(raw_out, np.float64, n_rows * t_samples)[0] if raw_out is not None
              else _vp(None))
        _check(qmlib.qm_engine_onsets(
            self._h, ps, dev_s, n_traces, t_samples, trace_row, n_rows, nsta, nlta,
            {"energy": 0, "abs": 1}[transform],
            {"classic": 0, "centred": 1, "recursive": 2}[position],
            int(taper_pad), float(min_onset_value), pr, pl, dev_o))
        return raw_out, log_out

    def find_max_coa(self, map4d, n_samples, n_nodes, out=None):
        if out is None:
            out = (np.zeros(n_samples), np.zeros(n_samples),
                   np.zeros(n_samples, dtype=np.int64))
        pm, dev_map = self._ptr(map4d, np.float64, int(n_samples) * int(n_nodes))
        pa, pb, pc, da = self._series(out, int(n_samples))
        _check(qmlib.qm_engine_find_max_coa(self._h, pm, dev_map, int(n_samples),
                                            int(n_nodes), pa, pb, pc, da))
        return out

    def _check_rows(self, rows):
        if self.n_rows is None:
            raise QMHipError("no travel-time table resident: call load_lut first")
        if rows != self.n_rows:
            raise ValueError(
                "Mismatch between number of stations for data and LUT, "
                f"{rows}:{self.n_rows}")


def timeit(*args_, **kwargs_):
    """
    The reference's per-call wall-time log line (quakemigrate/util.py:651-669, applied at
    core/lib.py:52,131 and signal/scan.py:593): same message, ``timeit("info")`` logs at info
    level, ``timeit()`` at debug.  The wrapped functions here take host arrays and return host
    arrays, so the elapsed time includes the copies and the wait for the GPU.
    """
    import functools
    import time

    def inner_function(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            ts = time.time()
            result = func(*args, **kwargs)
            msg = " " * 21 + f"Elapsed time: {time.time() - ts:6f} seconds."
            if args_ and args_[0] == "info":
                logging.info(msg)
            else:
                logging.debug(msg)
            return result

        return wrapper

    return inner_function


# --------------------------------------------------------------------------
# module-level engine for the reference-signature functions.  They receive the
# table on every call (lib.py:53-60); like the C symbols beside them
# (qm_compat.hip: migrate) they keep it resident and upload it again only when
# its shape or content -- two independent 64-bit hashes of every word,
# qm_table_hash -- changes (QM_HIP_COMPAT_REUPLOAD=1: on every call).  Callers
# that keep a table across timesteps use an Engine (or
# quakemigrate_amd.scan.MigrationScan) and never pay for the hash either.
# --------------------------------------------------------------------------
_default = {"engine": None, "table_key": None}


def release_cached_memory():
    """Return the device memory that destroyed engines / replaced tables left parked in the process
    (``qm_release_cached_memory``) to the driver."""
    qmlib.qm_release_cached_memory()


def default_engine():
    if _default["engine"] is None:
        import os

        _default["engine"] = Engine(int(os.environ.get("QM_HIP_DEVICE", "0")))
    return _default["engine"]


def _table_key(traveltimes):
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    qmlib.qm_table_hash(traveltimes.ctypes.data_as(ctypes.c_void_p), traveltimes.size,
                        ctypes.byref(a), ctypes.byref(b))
    return (tuple(traveltimes.shape), int(a.value), int(b.value))


def _resident(traveltimes):
    import os

    eng = default_engine()
    key = _table_key(traveltimes)
    if (os.environ.get("QM_HIP_COMPAT_REUPLOAD", "0") not in ("", "0")
            or _default["table_key"] != (key, eng.table_generation)):
        _default["table_key"] = None
        eng.load_lut(traveltimes)
        _default["table_key"] = (key, eng.table_generation)
    return eng


def _prepare(onsets, traveltimes, first_idx, last_idx):
    """Pre-processing and checks of lib.py:93-110, in the reference's order."""
    onsets = np.clip(onsets, 0.01, np.inf)
    onsets = np.ascontiguousarray(np.log(onsets), dtype=np.float64)
    *grid_dimensions, n_luts = traveltimes.shape
    n_onsets, t_samples = onsets.shape
    logging.debug(f"(n_onsets, t_samples) : ({n_onsets}, {t_samples})")
    n_samples = t_samples - first_idx - last_idx
    logging.debug(f"n_samples : {n_samples}")
    if not n_luts == n_onsets:
        raise ValueError(
            f"Mismatch between number of stations for data and LUT, {n_onsets}:{n_luts}"
        )
    if onsets.size < n_samples + first_idx:
        raise ValueError("Data array smaller than coalescence array.")
    if traveltimes.dtype != np.int32 or not traveltimes.flags["C_CONTIGUOUS"]:
        # the reference's ndpointer argtype raises ctypes.ArgumentError here
        raise ctypes.ArgumentError(
            "traveltimes must be a C-contiguous int32 array")
    return onsets, tuple(grid_dimensions), n_samples


@timeit()
def migrate(onsets, traveltimes, first_idx, last_idx, available, threads=1):
    """
    Computes 4-D coalescence map by migrating seismic phase onset functions.

    Same contract as the reference's ``migrate`` (lib.py:52-125); ``threads`` is
    accepted and ignored (the GPU engine has no thread knob).

    Returns
    -------
    map4d: 4-D coalescence map, shape(nx, ny, nz, nsamples).
    """
    onsets, grid, n_samples = _prepare(onsets, traveltimes, first_idx, last_idx)
    eng = _resident(traveltimes)
    map4d = np.zeros(grid + (n_samples,), dtype=np.double)
    logging.debug(f"map4d shape : {map4d.shape}")
    eng.migrate(onsets, first_idx, last_idx, available, map4d)
    return map4d


@timeit()
def find_max_coa(map4d, threads=1):
    """
    Finds time series of the maximum coalescence/normalised coalescence in the
    3-D volume, and the corresponding grid indices (reference lib.py:131-170).
    """
    *grid_dimensions, n_samples = map4d.shape
    n_nodes = int(np.prod(grid_dimensions))
    max_coa = np.zeros(n_samples, dtype=np.double)
    max_norm_coa = np.zeros(n_samples, dtype=np.double)
    max_coa_idx = np.zeros(n_samples, dtype=np.int64)
    default_engine().find_max_coa(
        np.ascontiguousarray(map4d, dtype=np.double), n_samples, n_nodes,
        (max_coa, max_norm_coa, max_coa_idx))
    return max_coa, max_norm_coa, max_coa_idx


@timeit()
def migrate_and_find_max(onsets, traveltimes, first_idx, last_idx, available,
                         threads=1, return_map=False):
    """
    Fused ``migrate`` + ``find_max_coa`` (what ``QuakeScan._compute`` does back to
    back, scan.py:635-638).  With ``return_map=False`` (detect) the 4-D map is
    never written anywhere.

    Returns ``(max_coa, max_norm_coa, max_coa_idx)`` or, with ``return_map``,
    ``(max_coa, max_norm_coa, max_coa_idx, map4d)``.
    """
    onsets, grid, n_samples = _prepare(onsets, traveltimes, first_idx, last_idx)
    eng = _resident(traveltimes)
    out = (np.zeros(n_samples), np.zeros(n_samples),
           np.zeros(n_samples, dtype=np.int64))
    if not return_map:
        eng.detect(onsets, first_idx, last_idx, available, out=out)
        return out
    map4d = np.zeros(grid + (n_samples,), dtype=np.double)
    eng.migrate(onsets, first_idx, last_idx, available, map4d, scan_out=out)
    return out + (map4d,)


def _stalta(fn, signal, nsta, nlta, fill):
    head = np.empty(1, dtype=stalta_header_t)
    head[:] = (len(signal), nsta, nlta)
    signal = np.ascontiguousarray(signal, dtype=np.double)
    onset = np.full(len(signal), fill, dtype=np.double)
    fn(signal, head, onset)
    return onset


def overlapping_sta_lta(signal, nsta, nlta):
    """Overlapping-window STA/LTA (reference lib.py:176-208)."""
    return _stalta(qmlib.overlapping_sta_lta, signal, nsta, nlta, 1.0)


def centred_sta_lta(signal, nsta, nlta):
    """Centred STA/LTA (reference lib.py:214-246)."""
    return _stalta(qmlib.centred_sta_lta, signal, nsta, nlta, 1.0)


def recursive_sta_lta(signal, nsta, nlta):
    """Recursive STA/LTA (reference lib.py:252-285)."""
    return _stalta(qmlib.recursive_sta_lta, signal, nsta, nlta, 0.0)
