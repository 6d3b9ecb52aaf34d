# -*- coding: utf-8 -*-
"""
``.scanmseed`` without obspy: the wire format ``QuakeScan.detect()`` emits and ``Trigger`` reads
(quakemigrate/io/scanmseed.py:74-150, 222-240, 244-325).

The reference hands five int32 series -- COA*1e5, COA_N*1e5, X*1e6, Y*1e6, Z*1e3*ucf, the two
coalescence series clipped at 21474 first -- to obspy, which writes big-endian miniSEED 2 with
4096-byte records, blockette 1000, STEIM2 compression (encoding 11), one trace after the other,
station codes COA, COA_N, X, Y, Z, network NW.  This module quantises exactly like
``ScanmSEED._data2int`` and reads / writes that container directly, so the engine's detect
output can be compared with the reference's golden files
(``examples/benchmarks/*/2014_*.scanmseed``) and fed to the reference's ``Trigger`` unchanged.

STEIM2 (SEED manual, appendix B): 64-byte frames of 16 big-endian words; word 0 holds sixteen
2-bit codes; in the first frame words 1 and 2 are the forward and reverse integration constants
(first and last sample of the record).  Codes: 00 nothing, 01 four 8-bit differences, 10 and 11
select by the word's top two bits ("dnib") 1x30 / 2x15 / 3x10 and 5x6 / 6x5 / 7x4 differences.
The encoder packs greedily -- the widest count whose differences all fit -- as libmseed (the
library behind obspy's writer) does, which reproduces the reference's files byte for byte
(tests/test_scanmseed.py).
"""

from __future__ import annotations

import datetime as _dt
import logging
import struct
from dataclasses import dataclass

import numpy as np

RECLEN = 4096
DATA_OFFSET = 64
N_FRAMES = (RECLEN - DATA_OFFSET) // 64
CHANNELS = ("COA", "COA_N", "X", "Y", "Z")
COA_CLIP = 21474.0


def scale_factors(ucf=1.0):
    """Multipliers of the five channels (scanmseed.py:113-130)."""
    return {"COA": 1e5, "COA_N": 1e5, "X": 1e6, "Y": 1e6, "Z": 1e3 * ucf}


def data2int(data, factor):
    """``ScanmSEED._data2int`` (scanmseed.py:222-240): round half to even, int32."""
    return np.round(np.asarray(data, dtype=np.float64) * factor).astype(np.int32)


def quantise(max_coa, max_coa_n, coord, ucf=1.0):
    """The five int32 series of one ``append`` call (scanmseed.py:103-130)."""
    max_coa = np.minimum(np.asarray(max_coa, dtype=np.float64), COA_CLIP)
    max_coa_n = np.minimum(np.asarray(max_coa_n, dtype=np.float64), COA_CLIP)
    f = scale_factors(ucf)
    coord = np.asarray(coord, dtype=np.float64)
    return {"COA": data2int(max_coa, f["COA"]), "COA_N": data2int(max_coa_n, f["COA_N"]),
            "X": data2int(coord[:, 0], f["X"]), "Y": data2int(coord[:, 1], f["Y"]),
            "Z": data2int(coord[:, 2], f["Z"])}


# --------------------------------------------------------------------------------------
# STEIM2
# --------------------------------------------------------------------------------------
def _sign_extend(v, bits):
    return v - (1 << bits) if v & (1 << (bits - 1)) else v


def steim2_decode(payload: bytes, n_samples: int) -> np.ndarray:
    """Decode the data part of one record; checks the reverse integration constant."""
    words = struct.unpack(f">{len(payload) // 4}I", payload)
    x0 = _sign_extend(words[1], 32)
    xn = _sign_extend(words[2], 32)
    diffs = []
    for frame in range(len(words) // 16):
        w = words[frame * 16: frame * 16 + 16]
        for k in range(1, 16):
            if frame == 0 and k < 3:
                continue
            code = (w[0] >> (30 - 2 * k)) & 3
            v = w[k]
            if code == 0:
                continue
            if code == 1:
                layout = (4, 8)
            else:
                dnib = v >> 30
                layout = {(2, 1): (1, 30), (2, 2): (2, 15), (2, 3): (3, 10),
                          (3, 0): (5, 6), (3, 1): (6, 5), (3, 2): (7, 4)}.get((code, dnib))
                if layout is None:
                    raise ValueError(f"invalid STEIM2 code {code} / dnib {dnib}")
            n, bits = layout
            for i in range(n):
                shift = (n - 1 - i) * bits
                diffs.append(_sign_extend((v >> shift) & ((1 << bits) - 1), bits))
            if len(diffs) >= n_samples:
                break
        if len(diffs) >= n_samples:
            break
    if len(diffs) < n_samples:
        raise ValueError("record holds fewer differences than samples")
    out = np.empty(n_samples, dtype=np.int64)
    out[0] = x0
    if n_samples > 1:
        out[1:] = x0 + np.cumsum(np.array(diffs[1:n_samples], dtype=np.int64))
    if out[-1] != xn:
        raise ValueError("STEIM2 reverse integration constant mismatch")
    return out.astype(np.int32)


def _fits(v, bits):
    return -(1 << (bits - 1)) <= v <= (1 << (bits - 1)) - 1


_LAYOUTS = ((7, 4, 3, 2), (6, 5, 3, 1), (5, 6, 3, 0), (4, 8, 1, None),
            (3, 10, 2, 3), (2, 15, 2, 2), (1, 30, 2, 1))   # (count, bits, code, dnib)


def steim2_encode(samples: np.ndarray, prev: int, max_frames: int = N_FRAMES):
    """
    Pack as many of ``samples`` as fit into ``max_frames`` frames.  ``prev`` is the sample before
    the first one (difference 0 of the record).  Returns ``(payload bytes, n_packed)``.
    """
    x = [int(v) for v in samples]
    diffs = [x[0] - int(prev)] + [x[i] - x[i - 1] for i in range(1, len(x))]
    frames = [[0] * 16 for _ in range(max_frames)]
    pos = 0                                            # next difference to pack
    for fi in range(max_frames):
        frame = frames[fi]
        for k in range(1, 16):
            if fi == 0 and k < 3:
                continue
            if pos >= len(diffs):
                break
            left = len(diffs) - pos
            for count, bits, code, dnib in _LAYOUTS:
                if count <= left and all(_fits(d, bits) for d in diffs[pos:pos + count]):
                    break
            else:
                raise OverflowError("difference does not fit 30 bits: not STEIM2-encodable")
            word = 0
            for d in diffs[pos:pos + count]:
                word = (word << bits) | (d & ((1 << bits) - 1))
            if dnib is not None:
                word |= dnib << 30
            frame[k] = word
            frame[0] |= code << (30 - 2 * k)
            pos += count
        if pos >= len(diffs):
            break
    frames[0][1] = x[0] & 0xFFFFFFFF
    frames[0][2] = x[pos - 1] & 0xFFFFFFFF
    payload = b"".join(struct.pack(">16I", *f) for f in frames)
    return payload, pos


# --------------------------------------------------------------------------------------
# miniSEED 2 records
# --------------------------------------------------------------------------------------
@dataclass
class Trace:
    station: str
    network: str
    starttime: _dt.datetime           # UTC, first sample
    sampling_rate: float
    data: np.ndarray                  # int32

    @property
    def endtime(self):
        return self.starttime + _dt.timedelta(seconds=(len(self.data) - 1) / self.sampling_rate)


def _btime(t: _dt.datetime) -> bytes:
    frac = int(round(t.microsecond / 100.0))
    return struct.pack(">HHBBBBH", t.year, t.timetuple().tm_yday, t.hour, t.minute, t.second,
                       0, frac)


def _from_btime(raw: bytes) -> _dt.datetime:
    year, doy, hh, mm, ss, _, frac = struct.unpack(">HHBBBBH", raw)
    return (_dt.datetime(year, 1, 1) + _dt.timedelta(days=doy - 1, hours=hh, minutes=mm,
                                                      seconds=ss, microseconds=frac * 100))


def _rate_fields(rate: float):
    if rate >= 1 and float(rate).is_integer():
        return int(rate), 1
    raise ValueError("only integer sampling rates (samples per second) are supported")


def write_trace(trace: Trace) -> bytes:
    """All records of one trace (what obspy writes for one ``Trace`` with STEIM2, reclen 4096)."""
    out = []
    data = np.asarray(trace.data, dtype=np.int32)
    pos, seq = 0, 1
    factor, mult = _rate_fields(trace.sampling_rate)
    while pos < len(data):
        prev = int(data[pos - 1]) if pos else int(data[0])     # first record: difference 0 = 0
        payload, n = steim2_encode(data[pos:], prev)
        start = trace.starttime + _dt.timedelta(seconds=pos / trace.sampling_rate)
        header = (f"{seq:06d}".encode() + b"D " + trace.station.ljust(5).encode() + b"  "
                  + b"   " + trace.network.ljust(2).encode() + _btime(start)
                  + struct.pack(">HhhBBBBiHH", n, factor, mult, 0, 0, 0, 1, 0, DATA_OFFSET, 48)
                  + struct.pack(">HHBBBB", 1000, 0, 11, 1, 12, 0))
        out.append(header.ljust(DATA_OFFSET, b"\x00") + payload)
        pos += n
        seq += 1
    return b"".join(out)


def write_scanmseed(path, starttime, sampling_rate, series, network="NW"):
    """``series``: dict channel -> int32 array (see :func:`quantise`), written in CHANNELS order."""
    blob = b"".join(write_trace(Trace(ch, network, starttime, sampling_rate, series[ch]))
                    for ch in CHANNELS)
    with open(path, "wb") as f:
        f.write(blob)
    return len(blob)


def read_records(blob: bytes):
    """Yield ``(station, network, starttime, sampling_rate, int32 data)`` per record."""
    for off in range(0, len(blob), RECLEN):
        rec = blob[off:off + RECLEN]
        station = rec[8:13].decode().strip()
        network = rec[18:20].decode().strip()
        start = _from_btime(rec[20:30])
        n, factor, mult, _, _, _, nblk, _, dstart, bstart = struct.unpack(">HhhBBBBiHH", rec[30:48])
        encoding = None
        b = bstart
        for _ in range(nblk):
            btype, nxt = struct.unpack(">HH", rec[b:b + 4])
            if btype == 1000:
                encoding, order, reclen_exp = rec[b + 4], rec[b + 5], rec[b + 6]
                if order != 1 or (1 << reclen_exp) != RECLEN:
                    raise ValueError("only big-endian 4096-byte records are supported")
            b = nxt
        if encoding != 11:
            raise ValueError(f"encoding {encoding}: only STEIM2 (11) is supported")
        rate = float(factor) * (mult if mult > 0 else -1.0 / mult) if factor > 0 else -1.0 / factor
        yield station, network, start, rate, steim2_decode(rec[dstart:], n)


def read_scanmseed(path, ucf=1.0):
    """
    Read one ``.scanmseed`` file: returns ``(starttime, sampling_rate, columns)`` with
    ``columns`` = dict of float64 arrays COA, COA_N, X, Y, Z, descaled as
    ``read_scanmseed`` does (scanmseed.py:300-305), plus the raw int32 series under ``"int"``.
    """
    with open(path, "rb") as f:
        blob = f.read()
    pieces = {}
    start, rate = {}, None
    for station, _, t0, sr, data in read_records(blob):
        pieces.setdefault(station, []).append((t0, data))
        rate = sr
        start[station] = min(start.get(station, t0), t0)
    ints = {}
    for station, recs in pieces.items():
        recs.sort(key=lambda r: r[0])
        # records of a trace must be contiguous in time
        t = recs[0][0]
        for t0, d in recs:
            if abs((t0 - t).total_seconds()) > 0.5 / rate:
                raise ValueError(f"gap in {station} at {t0}")
            t = t0 + _dt.timedelta(seconds=len(d) / rate)
        ints[station] = np.concatenate([d for _, d in recs])
    f = scale_factors(ucf)
    cols = {ch: ints[ch] / f[ch] for ch in CHANNELS}
    cols["int"] = ints
    return start[CHANNELS[0]], rate, cols


# --------------------------------------------------------------------------------------
# The detect loop's sink
# --------------------------------------------------------------------------------------
class CoalescenceSink:
    """
    The reference's ``ScanmSEED`` object (quakemigrate/io/scanmseed.py:27-220) without obspy, as far
    as the detect loop uses it (``QuakeScan._continuous_compute``, signal/scan.py:421-466):
    ``append`` clips, quantises and appends one timestep to the five int32 channels
    (scanmseed.py:74-131) and writes a day's file when the stream reaches the day line or crosses it
    (scanmseed.py:133-150: the part before midnight is written, the rest kept); ``empty`` appends an
    all-zero timestep (scanmseed.py:152-180); ``write`` writes what is held (one
    ``<year>_<julday>.scanmseed`` per day, scanmseed.py:182-220).  Times are ``datetime`` (UTC).
    """

    def __init__(self, directory, sampling_rate, continuous_write=False):
        import pathlib

        self.directory = pathlib.Path(directory)
        self.sampling_rate = float(sampling_rate)
        self.continuous_write = continuous_write
        self.written = False
        self.starttime = None
        self.series = {ch: np.zeros(0, dtype=np.int32) for ch in CHANNELS}
        self.files = []

    def _endtime(self):
        n = len(self.series["COA"])
        return self.starttime + _dt.timedelta(seconds=(n - 1) / self.sampling_rate)

    def append(self, starttime, max_coa, max_coa_n, coord, ucf):
        new = quantise(max_coa, max_coa_n, coord, ucf)
        if self.starttime is None or len(self.series["COA"]) == 0:
            self.starttime = starttime
        else:
            expect = self._endtime() + _dt.timedelta(seconds=1.0 / self.sampling_rate)
            if abs((starttime - expect).total_seconds()) > 0.5 / self.sampling_rate:
                raise ValueError(f"timestep starting {starttime} does not continue the stream ending "
                                 f"{self._endtime()}")
        for ch in CHANNELS:
            self.series[ch] = np.concatenate([self.series[ch], new[ch]])
        self.written = False
        delta = _dt.timedelta(seconds=1.0 / self.sampling_rate)
        start, end = self.starttime, self._endtime()
        midnight = _dt.datetime(start.year, start.month, start.day) + _dt.timedelta(days=1)
        if end == midnight - delta:                       # passed the day line: write, start afresh
            self.write()
            self.starttime = None
            self.series = {ch: np.zeros(0, dtype=np.int32) for ch in CHANNELS}
        elif end >= midnight:
            logging.debug("Timestep doesn't fall at midnight!")
            keep = int(round((midnight - start).total_seconds() * self.sampling_rate))
            rest = {ch: self.series[ch][keep:] for ch in CHANNELS}
            self.series = {ch: self.series[ch][:keep] for ch in CHANNELS}
            self.write()
            self.starttime, self.series = midnight, rest
            self.written = False                          # (residual data not yet in a file)
        if self.continuous_write and not self.written:
            self.write()

    def empty(self, starttime, timestep, i, msg, ucf):
        logging.info(msg)
        n = int(round(timestep * int(self.sampling_rate)))            # util.time2sample
        start = starttime + _dt.timedelta(seconds=timestep * i)
        self.append(start, np.zeros(n), np.zeros(n), np.zeros((n, 3)), ucf)

    def write(self, write_start=None, write_end=None):
        if self.starttime is None or len(self.series["COA"]) == 0:
            self.written = True
            return
        self.directory.mkdir(parents=True, exist_ok=True)
        t = self.starttime
        path = self.directory / f"{t.year}_{t.timetuple().tm_yday:03d}.scanmseed"
        write_scanmseed(path, t, self.sampling_rate, self.series)
        if path not in self.files:
            self.files.append(path)
        self.written = True
