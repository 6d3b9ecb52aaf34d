# -*- coding: utf-8 -*-
"""
Host-side counterpart of ``QuakeScan._compute`` for the migrate / find_max_coa path.

``MigrationScan.continuous_compute(...)`` is the loop around it (reference ``scan.py:407-470``).
``MigrationScan._compute(data, event=None)`` reproduces the glue of the reference's
``quakemigrate/signal/scan.py:593-647`` around the hot path -- onset plugin ->
served travel-time table -> ``fsmp`` / ``lsmp`` -> ``available`` -> migrate ->
find_max_coa -> ``index2coord`` -- with the same argument meaning, return tuples and
error behaviour, but:

* the table is made resident on the GPU once per ``(sampling_rate, availability)``
  (the reference rebuilds and re-passes it every timestep, ``lut.py:502-538``), and
* in the detect stage the fused kernel runs, so the 4-D map is never materialised
  (the reference allocates it only to reduce and delete it, ``scan.py:641-642``).

It is duck-typed against the reference's plugin API, so the reference's own objects
drop in unchanged:

* ``onset``  : ``calculate_onsets(data) -> (onsets[S, T] float64, onset_data)`` with
  ``onset_data.sampling_rate`` and ``onset_data.availability`` (dict "STATION_PHASE"
  -> 0/1), see ``quakemigrate/signal/onsets/base.py:24-130, 133-191``;
* ``lut``    : ``serve_traveltimes(sampling_rate, availability) -> int32
  (nx, ny, nz, S)`` and ``index2coord(idx, unravel=True)``
  (``quakemigrate/lut/lut.py:502-538, 211-243``).

obspy / pyproj are not needed here: whatever ``data.starttime`` and
``event.mw_times`` return is passed through.
"""

from __future__ import annotations

import itertools
import logging

import numpy as np

from quakemigrate_amd.core import lib


_LUT_TOKENS = itertools.count(1)


def _lut_token(lut):
    """A process-unique name for ``lut`` in the keys of parked tables (``id()`` would be reused by a
    later object at the same address and bring a dead LUT's table back): kept on the object where
    it takes attributes, else the caller gets a fresh one (no sharing between scans)."""
    token = getattr(lut, "_qm_hip_token", None)
    if token is None:
        token = next(_LUT_TOKENS)
        try:
            lut._qm_hip_token = token
        except (AttributeError, TypeError):
            pass
    return token


class LUTPhasesException(Exception):
    """Mirror of ``quakemigrate.util.LUTPhasesException`` (util.py)."""


class _NoDataException(Exception):
    def __init__(self, msg="no data"):
        super().__init__(msg)
        self.msg = msg


class ArchiveEmptyException(_NoDataException):
    """Mirror of ``quakemigrate.util.ArchiveEmptyException`` (carries ``.msg``)."""


class DataGapException(_NoDataException):
    """Mirror of ``quakemigrate.util.DataGapException``."""


class DataAvailabilityException(_NoDataException):
    """Mirror of ``quakemigrate.util.DataAvailabilityException``."""


# the reference's own exception objects (raised by ITS archive / onset classes when both packages are
# installed) are recognised by name: nothing of the reference is imported here
_NO_DATA = ("ArchiveEmptyException", "DataGapException", "DataAvailabilityException")


def _is_no_data(exc):
    """The reference catches util.ArchiveEmptyException / DataGapException / DataAvailabilityException AND their
    subclasses (scan.py:449-458); its exception objects are recognised by the names in the raised class's MRO
    (obspy-free: the reference's util module is not imported here)."""
    return any(cls.__name__ in _NO_DATA for cls in type(exc).__mro__)


def _shift(t, seconds):
    """``t + seconds`` for obspy ``UTCDateTime`` (adds floats) and ``datetime`` alike."""
    try:
        return t + seconds
    except TypeError:
        import datetime as _dt

        return t + _dt.timedelta(seconds=float(seconds))


def time2sample(time, sampling_rate):
    """Seconds -> whole samples, as ``quakemigrate/util.py:152-172``."""
    return int(round(time * int(sampling_rate)))


class MigrationScan:
    """
    Parameters
    ----------
    lut, onset : plugin objects (see module docstring).
    pre_pad, post_pad : float
        Seconds of onset data before / after the scanned window (what
        ``QuakeScan`` gets from ``onset.pad(timestep)``, scan.py:425).
    stage : {"detect", "locate"}
        ``run.stage`` of the reference (scan.py:641).
    scan_rate : int, optional
        Passed to ``event.mw_times`` in the locate stage (scan.py:646).
    engine : quakemigrate_amd.core.Engine, optional
        Defaults to the process-wide engine on ``$QM_HIP_DEVICE``.
    threads : int
        Accepted for signature compatibility; the GPU engine ignores it.
    device_serving : bool
        Build the int32 table on the GPU from ``lut.traveltimes`` (see ``_ensure_table``).
    screen : bool, optional
        Detect precision, see ``INTEGRATION.md`` section 4b.
    table_cache : int
        Tables of other station availabilities kept parked on the device (``_ensure_table``);
        0 = one resident table, rebuilt at every change.
    """

    def __init__(self, lut, onset, pre_pad, post_pad, stage="detect", scan_rate=None,
                 engine=None, threads=1, device_serving=False, screen=None, table_cache=4):
        self.lut = lut
        self.onset = onset
        self.pre_pad = pre_pad
        self.post_pad = post_pad
        self.stage = stage
        self.scan_rate = scan_rate
        self.threads = threads
        self.engine = engine if engine is not None else lib.default_engine()
        # screen: None leaves the engine as configured (float64 throughout unless it was created
        # with screen=1); True / False select the opt-in screened detect (exact argmax and
        # max_coa, max_coa_n within 6.7e-7 by a deterministic bound) for THIS scan's steps only:
        # the setting is applied around each call and the engine's own restored afterwards (the
        # default engine is shared by every MigrationScan and by lib.migrate_and_find_max)
        self.screen = None if screen is None else bool(screen)
        self._lut_token = _lut_token(lut)
        self._resident_key = None
        self._resident_generation = -1
        self._grids_generation = -1
        self.table_cache = int(table_cache)
        # device_serving: the float64 grids of ``lut.traveltimes`` ({station: {phase: grid}},
        # quakemigrate/lut/lut.py) are uploaded once and the int32 table of the available
        # station/phase pairs is built on the GPU (rint(tt * sampling_rate), lut.py:536-538)
        # whenever the availability changes -- no per-timestep host stack / rint / upload.
        self.device_serving = bool(device_serving)
        self._grid_index = None

    # -- table residency ------------------------------------------------------
    def _ensure_table(self, sampling_rate, availability):
        """
        Make the table of this ``(sampling_rate, availability)`` the engine's resident one.  The
        reference serves it anew every timestep (lut.py:502-538); here it is built the first time
        an availability is seen and PARKED on the device when another one takes over
        (``Engine.select_table``, up to ``table_cache`` tables, least recently used evicted), so
        stations dropping in and out alternate between resident tables at no cost.
        """
        key = (sampling_rate, tuple(availability.items()))
        eng = self.engine
        # (the engine may be shared -- the default one is -- so residency is the engine's word:
        # its table generation moves whenever anybody replaces or switches the table)
        if key == self._resident_key and eng.table_generation == self._resident_generation:
            return eng
        if eng.select_table((self._lut_token, key), capacity=self.table_cache):
            self._resident_key, self._resident_generation = key, eng.table_generation
            return eng
        if self.device_serving:
            if self._grid_index is None or self._grids_generation != getattr(eng, "grids_generation", 0):
                names, grids = [], []
                for station, phases in self.lut.traveltimes.items():
                    for phase, grid in phases.items():
                        names.append(f"{station}_{phase}")
                        grids.append(grid)
                eng.set_traveltime_grids(grids)
                self._grids_generation = eng.grids_generation
                self._grid_index = {n: i for i, n in enumerate(names)}
            rows = []
            for k, available in availability.items():
                if available != 1:
                    continue
                station, phase = k.split("_")
                if k in self._grid_index:
                    rows.append(self._grid_index[k])
                elif f"{station}_TIME_{phase}" in self._grid_index:      # lut.py:534-535
                    rows.append(self._grid_index[f"{station}_TIME_{phase}"])
                else:
                    phases = sorted({kk.split("_")[-1] for kk in availability})
                    raise LUTPhasesException(
                        f"Attempting to migrate phases {phases}; but traveltimes for "
                        f"'{phase}' not found in the LUT. Please create a new lookup table "
                        f"with phases={phases}")
            eng.serve(sampling_rate, rows)
        else:
            try:
                traveltimes = self.lut.serve_traveltimes(sampling_rate, availability)
            except KeyError as e:
                phases = sorted({k.split("_")[-1] for k in availability})
                raise LUTPhasesException(
                    f"Attempting to migrate phases {phases}; but traveltimes for {e} "
                    f"not found in the LUT. Please create a new lookup table with "
                    f"phases={phases}")
            traveltimes = np.ascontiguousarray(traveltimes, dtype=np.int32)
            eng.load_lut(traveltimes)
            logging.debug("travel-time table %s made resident", traveltimes.shape)
        self._resident_key, self._resident_generation = key, eng.table_generation
        return eng

    # -- the hot-path glue ------------------------------------------------------
    @lib.timeit("info")
    def _compute(self, data, event=None):
        """
        Compute 3-D coalescence between two time stamps (reference scan.py:593-647).

        Returns (detect) ``time, max_coa, max_coa_n, coord, onset_data`` or
        (locate) ``times, max_coa, max_coa_n, coord, map4d, onset_data``.
        """
        onsets, onset_data = self.onset.calculate_onsets(data)
        eng = self._ensure_table(onset_data.sampling_rate, onset_data.availability)
        fsmp = time2sample(self.pre_pad, onset_data.sampling_rate)
        lsmp = time2sample(self.post_pad, onset_data.sampling_rate)
        avail = int(np.sum([value for _, value in onset_data.availability.items()]))

        # lib.py:93-110 -- same pre-processing and checks, same order
        onsets = np.ascontiguousarray(np.log(np.clip(onsets, 0.01, np.inf)))
        n_onsets, t_samples = onsets.shape
        n_samples = t_samples - fsmp - lsmp
        if n_onsets != eng.n_rows:
            raise ValueError("Mismatch between number of stations for data and LUT, "
                             f"{n_onsets}:{eng.n_rows}")
        if onsets.size < n_samples + fsmp:
            raise ValueError("Data array smaller than coalescence array.")

        series = (np.zeros(n_samples), np.zeros(n_samples),
                  np.zeros(n_samples, dtype=np.int64))
        if self.stage == "detect":
            previous = eng.get("screen")
            if self.screen is not None:
                eng.config("screen", 1 if self.screen else 0)
            try:
                eng.detect(onsets, fsmp, lsmp, avail, out=series)
            finally:
                if self.screen is not None:
                    eng.config("screen", previous)
            map4d = None
        else:
            map4d = np.zeros(tuple(eng.grid) + (n_samples,), dtype=np.double)
            eng.migrate(onsets, fsmp, lsmp, avail, map4d, scan_out=series)
        max_coa, max_coa_n, max_idx = series
        coord = self.lut.index2coord(max_idx, unravel=True)

        if self.stage == "detect":
            time = data.starttime + self.pre_pad
            return time, max_coa, max_coa_n, coord, onset_data
        times = event.mw_times(self.scan_rate)
        return times, max_coa, max_coa_n, coord, map4d, onset_data


    # -- the loop around the path ------------------------------------------------------
    def continuous_compute(self, archive, starttime, n_steps, timestep, scan_rate, sink,
                           steps_per_launch=None, depth=3):
        """
        ``QuakeScan._continuous_compute`` (reference scan.py:407-470): coalescence between two
        timestamps in increments of ``timestep`` -- per timestep read the waveforms, compute the onsets,
        migrate, scan, hand ``(time, max_coa, max_coa_n, coord)`` to ``sink.append`` -- with the
        reference's behaviour around the path: a timestep whose data raise ``ArchiveEmptyException`` /
        ``DataGapException`` / ``DataAvailabilityException`` becomes an all-zero timestep
        (``sink.empty(starttime, timestep, i, e.msg, ucf)``, scan.py:449-458) with an all-zero
        availability row, and ``sink.write()`` closes the run if the last append did not.

        ``archive``: ``read_waveform_data(w_beg, w_end)`` (the reference's ``Archive``);
        ``sink``: the reference's ``ScanmSEED`` duck-typed (``append`` / ``empty`` / ``write`` /
        ``written``; ``quakemigrate_amd.scanmseed.CoalescenceSink`` is one without obspy).
        The window arithmetic is the reference's (``w_beg = starttime + timestep * i - pre_pad``,
        ``w_end = starttime + timestep * (i + 1) - 1 / scan_rate + post_pad``).

        What differs is only WHEN the GPU works: timesteps are independent given their onsets, so
        their log-onsets go through the native pipeline (``qm_stream_*``: H2D, one fused launch per
        ``steps_per_launch`` timesteps, D2H on their own streams) while the host reads and
        pre-processes the next timestep; results reach the sink in timestep order.  A change of
        station availability (another served table) or a data gap drains the pipeline first.

        ``steps_per_launch``: None = 8 on grids of up to a million nodes (where one timestep is a
        fraction of a millisecond and a launch's fixed costs show: x1.04-1.09 of the resident step
        instead of x1.3-1.4), else 1; results do not depend on it.

        Returns the availability rows (list of dicts, one per timestep; scan.py:428, 448, 458).
        """
        from collections import deque

        if self.stage != "detect":
            raise ValueError("continuous_compute is the detect stage's loop")
        ucf = getattr(self.lut, "unit_conversion_factor", 1.0)
        rows = []
        pending = deque()                  # timesteps in the pipeline: (time, onset_data), oldest first
        state = {"stream": None, "key": None}

        def emit(n):
            a, b, c = state["stream"].pop(n)
            for j in range(n):
                time, onset_data = pending.popleft()
                coord = self.lut.index2coord(c[j], unravel=True)
                logging.debug(f"1-D con shape : {a[j].shape}")
                sink.append(time, a[j], b[j], coord, ucf)

        def drain():
            if state["stream"] is not None:
                state["stream"].flush()
                while pending:
                    emit(min(state["stream"].k, len(pending)))

        # (what the pipeline holds when anything else goes wrong mid-run is sunk before the error travels on: the
        # reference has appended every timestep it computed, scan.py:434-448)
        try:
            self._continuous_steps(archive, starttime, timestep, scan_rate, n_steps, sink, ucf, rows, pending,
                                   state, emit, drain, steps_per_launch, depth)
        except BaseException:
            try:
                drain()
            except Exception:  # noqa: BLE001  (the first error is the one to report)
                pass
            raise
        finally:
            if state["stream"] is not None:
                state["stream"].close()
            self._restore_screen(state.get("engine"), state.get("screen_before"))
        if not getattr(sink, "written", False):
            sink.write()
        columns = next((list(r) for r in rows if r is not None), [])
        return [r if r is not None else dict.fromkeys(columns, 0) for r in rows]

    def _apply_screen(self, eng):
        """MigrationScan.screen on the engine for the calls that follow (as _compute does per step); returns what
        to restore."""
        previous = eng.get("screen")
        if self.screen is not None:
            eng.config("screen", 1 if self.screen else 0)
        return previous

    def _restore_screen(self, eng, previous):
        if eng is not None and previous is not None and self.screen is not None:
            eng.config("screen", previous)

    def _continuous_steps(self, archive, starttime, timestep, scan_rate, n_steps, sink, ucf, rows, pending, state,
                          emit, drain, steps_per_launch, depth):
        from quakemigrate_amd.stream import StreamingDetector

        for i in range(n_steps):
            w_beg = _shift(_shift(starttime, timestep * i), -self.pre_pad)
            w_end = _shift(_shift(starttime, timestep * (i + 1) - 1 / scan_rate), self.post_pad)
            logging.debug(f" Processing : {w_beg}-{w_end} ".center(110, "~"))
            try:
                data = archive.read_waveform_data(w_beg, w_end)
                onsets, onset_data = self.onset.calculate_onsets(data)
            except Exception as e:  # noqa: BLE001
                if not _is_no_data(e):
                    raise
                drain()
                sink.empty(starttime, timestep, i, getattr(e, "msg", str(e)), ucf)
                rows.append(None)
                continue
            # (another availability = another table: what is in the pipeline belongs to the one
            # that is resident NOW and has to go through before the engine switches)
            if (onset_data.sampling_rate, tuple(onset_data.availability.items())) != self._resident_key:
                drain()
            eng = self._ensure_table(onset_data.sampling_rate, onset_data.availability)
            if state.get("engine") is None:              # (the scan's `screen` holds for the whole run)
                state["engine"], state["screen_before"] = eng, self._apply_screen(eng)
            fsmp = time2sample(self.pre_pad, onset_data.sampling_rate)
            lsmp = time2sample(self.post_pad, onset_data.sampling_rate)
            avail = int(np.sum([value for _, value in onset_data.availability.items()]))
            onsets = np.ascontiguousarray(np.log(np.clip(onsets, 0.01, np.inf)))      # lib.py:93-94
            n_onsets, t_samples = onsets.shape
            if n_onsets != eng.n_rows:
                raise ValueError("Mismatch between number of stations for data and LUT, "
                                 f"{n_onsets}:{eng.n_rows}")
            if onsets.size < t_samples - lsmp:
                raise ValueError("Data array smaller than coalescence array.")
            key = (self._resident_key, t_samples, fsmp, lsmp, avail)
            if key != state["key"]:                      # another table or window shape: a new pipeline
                drain()
                if state["stream"] is not None:
                    state["stream"].close()
                k = steps_per_launch if steps_per_launch else (8 if eng.n_nodes <= 1_000_000 else 1)
                state["stream"] = StreamingDetector(eng, n_onsets, t_samples, fsmp, lsmp, avail,
                                                    depth=depth, steps_per_launch=k)
                state["key"] = key
            stream = state["stream"]
            while not stream.push(onsets):
                emit(min(stream.k, stream.pending()[0]))
            pending.append((_shift(data.starttime, self.pre_pad), onset_data))
            rows.append(dict(onset_data.availability))
        drain()

    def locate_compute(self, archive, triggers, marginal_window, sgm=0.8, cov_thresh=0.90, on_event=None):
        """
        ``QuakeScan._locate_events``' loop around the path (reference scan.py:472-545), up to and including
        ``_calculate_location``: per triggered event read ``trigger_time -/+ (2 * marginal_window + pad)``
        (scan.py:497-498), compute the onsets, migrate and scan the ``4 * marginal_window * rate + 1``
        samples of ``event.mw_times`` (event.py:398-420); the origin time is the FIRST maximum of the
        coalescence series (``idxmax``, event.py:239-240); an event whose trigger time is not strictly
        inside ``otime -/+ marginal_window`` is dropped (``in_marginal_window``, event.py:369-396), as is
        one whose data raise ``ArchiveEmptyException`` / ``DataGapException`` / ``DataAvailabilityException``
        (scan.py:513-519); the series are trimmed to ``otime - mw <= t <= otime + mw`` and the 4-D map to the
        same samples WITHOUT the last one (``trim2window`` slices ``index[0]:index[-1]``, event.py:430-435:
        the reference's marginal sum runs over one sample fewer than its trimmed series has rows); the
        marginalised map is normalised and fitted three ways (scan.py:719-733).

        What differs is that the 4-D map never exists: the first pass is the fused detect kernel over the
        event's window (series only), which fixes the origin time; the second sums the marginal window's
        samples inside the stacking kernel (``calculate_location``) -- two launches of a few milliseconds
        on a C3-sized grid instead of a 13 GB volume.

        ``triggers``: iterable of ``(uid, trigger_time)``; time stamps only need ``t + timedelta`` /
        ``t + seconds`` (``_shift``).  ``on_event(result)`` is called per located event (the reference writes
        and picks there).  Returns the list of results, each a dict: ``uid, trigger_time, otime, times0``
        (time stamp of the first trimmed sample), ``first_sample, last_sample`` (trimmed series, inclusive,
        in scanned samples), ``max_coa, max_coa_n, coord`` (trimmed), ``coa_map`` (normalised), ``fits``
        (:class:`locate.LocationFits`), ``onset_data``.
        """
        from quakemigrate_amd import locate

        if self.stage != "locate":
            raise ValueError("locate_compute is the locate stage's loop")
        mw = float(marginal_window)
        results = []
        triggers = list(triggers)
        for n, (uid, trigger_time) in enumerate(triggers):
            w_beg = _shift(trigger_time, -2 * mw - self.pre_pad)
            w_end = _shift(trigger_time, 2 * mw + self.post_pad)
            logging.info(f"\tEVENT - {n + 1} of {len(triggers)} - {uid}")
            try:
                data = archive.read_waveform_data(w_beg, w_end)
                onsets, onset_data = self.onset.calculate_onsets(data)
            except Exception as e:  # noqa: BLE001
                if not _is_no_data(e):
                    raise
                logging.info(getattr(e, "msg", str(e)))
                continue
            rate = onset_data.sampling_rate
            eng = self._ensure_table(rate, onset_data.availability)
            fsmp, lsmp = time2sample(self.pre_pad, rate), time2sample(self.post_pad, rate)
            avail = int(np.sum([value for _, value in onset_data.availability.items()]))
            onsets = np.ascontiguousarray(np.log(np.clip(onsets, 0.01, np.inf)))      # lib.py:93-94
            n_onsets, t_samples = onsets.shape
            n_samples = t_samples - fsmp - lsmp
            if n_onsets != eng.n_rows:
                raise ValueError("Mismatch between number of stations for data and LUT, "
                                 f"{n_onsets}:{eng.n_rows}")
            if onsets.size < n_samples + fsmp:
                raise ValueError("Data array smaller than coalescence array.")
            # times[i] below takes the first scanned sample for trigger - 2 mw (event.py:412-420): the window read
            # must hold exactly 4 mw rate + 1 samples (the reference fails on the DataFrame's length otherwise)
            if n_samples != int(round(4 * mw * rate)) + 1:
                raise ValueError(f"Event {uid}: {n_samples} scanned samples where 4 * marginal_window * rate + 1 = "
                                 f"{int(round(4 * mw * rate)) + 1} are expected (pre_pad / post_pad / the archive's "
                                 "window do not match the marginal window)")
            series = (np.zeros(n_samples), np.zeros(n_samples), np.zeros(n_samples, dtype=np.int64))
            screen_before = self._apply_screen(eng)
            try:
                eng.detect(onsets, fsmp, lsmp, avail, out=series)                      # pass 1: the origin time
            finally:
                self._restore_screen(eng, screen_before)
            i_max = int(np.nanargmax(series[0]))                                       # idxmax: the first maximum
            # times[i] = trigger_time - 2 mw + i / rate (event.py:412-420)
            offset = i_max / rate - 2 * mw                                             # otime - trigger_time
            if not (-mw < offset < mw):                                                # event.py:381-383
                logging.info(f"\tEvent {uid} is outside marginal window.")
                continue
            eps = 1e-6                                                                 # (time stamps are whole ns)
            first = max(0, int(np.ceil(i_max - mw * rate - eps)))
            last = min(n_samples - 1, int(np.floor(i_max + mw * rate + eps)))
            if last <= first:
                raise ValueError(f"Event {uid}: the marginal window holds no sample interval (marginal_window "
                                 f"{mw} s at {rate} Hz: samples {first}..{last})")
            marginal = eng.marginal_map(onsets, fsmp, lsmp, avail, first, last)        # pass 2: samples [first, last)
            coa_map = np.zeros_like(marginal)
            fits = locate.calculate_location(eng, marginal, self.lut.node_spacing, sgm=sgm,
                                             cov_thresh=cov_thresh, norm_out=coa_map)
            sel = slice(first, last + 1)
            result = {
                "uid": uid, "trigger_time": trigger_time,
                "otime": _shift(trigger_time, offset), "times0": _shift(trigger_time, first / rate - 2 * mw),
                "first_sample": first, "last_sample": last,
                "max_coa": series[0][sel], "max_coa_n": series[1][sel],
                "coord": self.lut.index2coord(series[2][sel], unravel=True),
                "coa_map": coa_map, "fits": fits, "onset_data": onset_data,
            }
            results.append(result)
            if on_event is not None:
                on_event(result)
        return results

    def marginal_coalescence(self, data, first_sample, end_sample):
        """
        The marginalised 3-D coalescence map of one event window without the 4-D map: what
        ``event.trim2window()`` + ``np.sum(event.map4d, axis=-1)`` produce in the reference
        (quakemigrate/io/event.py:421-439, signal/scan.py:720), for the scanned samples
        ``[first_sample, end_sample)``.  Returns ``(coa_map (nx, ny, nz), max_coa, max_coa_n,
        max_idx, onset_data)``; the caller normalises by ``nanmax`` as the reference does.
        """
        onsets, onset_data = self.onset.calculate_onsets(data)
        eng = self._ensure_table(onset_data.sampling_rate, onset_data.availability)
        fsmp = time2sample(self.pre_pad, onset_data.sampling_rate)
        lsmp = time2sample(self.post_pad, onset_data.sampling_rate)
        avail = int(np.sum([value for _, value in onset_data.availability.items()]))
        onsets = np.ascontiguousarray(np.log(np.clip(onsets, 0.01, np.inf)))
        n_samples = onsets.shape[1] - fsmp - lsmp
        series = (np.zeros(n_samples), np.zeros(n_samples),
                  np.zeros(n_samples, dtype=np.int64))
        coa_map = eng.marginal_map(onsets, fsmp, lsmp, avail, first_sample, end_sample,
                                   scan_out=series)
        return (coa_map,) + series + (onset_data,)

    def calculate_location(self, data, first_sample, end_sample, sgm=0.8, cov_thresh=0.90):
        """
        ``QuakeScan._calculate_location`` (scan.py:696-733) for the marginal window
        ``[first_sample, end_sample)`` without the 4-D map ever existing: the marginalised map is
        summed inside the stacking kernel, normalised / smoothed / reduced on the GPU, and only
        the two fit windows come back for the Gaussian and spline algebra.

        Returns ``(coa_map, fits, max_coa, max_coa_n, max_idx, onset_data)``: ``coa_map`` is the
        normalised map the reference returns; ``fits`` a :class:`locate.LocationFits` (use
        ``fits.coordinates(self.lut)`` for the reference's lon / lat / depth triplets).
        """
        from quakemigrate_amd import locate

        marginal, max_coa, max_coa_n, max_idx, onset_data = self.marginal_coalescence(
            data, first_sample, end_sample)
        coa_map = np.zeros_like(marginal)
        fits = locate.calculate_location(self.engine, marginal, self.lut.node_spacing, sgm=sgm,
                                         cov_thresh=cov_thresh, norm_out=coa_map)
        return coa_map, fits, max_coa, max_coa_n, max_idx, onset_data
