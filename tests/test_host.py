# -*- coding: utf-8 -*-
"""
CPU-side checks of the boundary and the host logic (no GPU, no compute calls):

* the C-ABI library loads and exports every symbol include/qmhip.h declares;
* the binding-level error behaviour of quakemigrate/core/lib.py:105-110;
* the host STA/LTA symbols of the drop-in library against the reference vectors;
* x-plane sharding arithmetic;
* the cross-rank exchange (world_size 2, gloo) against the oracle's one-shot result.
"""

import ctypes
import os
import pathlib
import re
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build_engine()
    return g


def _declared_functions():
    text = (ROOT / "include" / "qmhip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(built):
    names = _declared_functions()
    assert {"migrate", "find_max_coa", "overlapping_sta_lta", "centred_sta_lta",
            "recursive_sta_lta", "qm_engine_create", "qm_engine_detect",
            "qm_engine_migrate", "qm_engine_finalize"} <= set(names)
    import sysconfig

    for libname in ("libqmhip.so", "qmlib" + sysconfig.get_config_var("EXT_SUFFIX")):
        lib = ctypes.CDLL(str(ROOT / "quakemigrate_amd" / "csrc" / libname))
        missing = [n for n in names if not hasattr(lib, n)]
        assert not missing, f"{libname} lacks {missing}"


def test_binding_imports_and_validates_like_the_reference(built):
    from quakemigrate_amd.core import lib

    on = np.ones((3, 50))
    with pytest.raises(ValueError, match="Mismatch"):
        lib.migrate(on, np.zeros((2, 2, 2, 4), dtype=np.int32), 2, 3, 3, 1)
    with pytest.raises(ValueError, match="smaller"):
        lib.migrate(np.ones((1, 4)), np.zeros((2, 2, 2, 1), dtype=np.int32), 8, -10, 1, 1)
    with pytest.raises(ctypes.ArgumentError):
        lib.migrate(on, np.zeros((2, 2, 2, 3), dtype=np.int64), 2, 3, 3, 1)
    assert lib.qmlib.qm_last_error() is not None


def test_drop_in_stalta_symbols_match_reference_vectors(built):
    from quakemigrate_amd.core import lib

    g = load_golden("stalta")
    toy = g["toy"]
    # reference tests/test_onsets.py:27-35
    assert (lib.overlapping_sta_lta(toy, 2, 3)
            == np.array([1.0, 1.0, 1.5, 1.25, 21.0 / 18, 27.0 / 24])).all()
    assert np.allclose(lib.centred_sta_lta(toy, 2, 3),
                       np.array([1.0, 1.0, 3.5, 2.25, 1.0, 1.0]))
    for kind in ("overlapping", "centred", "recursive"):
        fn = getattr(lib, f"{kind}_sta_lta")
        np.testing.assert_allclose(fn(toy, 2, 3), g[f"toy_{kind}"], rtol=1e-15)
        np.testing.assert_allclose(fn(g["signal"], int(g["nsta"]), int(g["nlta"])),
                                   g[kind], rtol=1e-12)


def test_engine_calls_fail_loudly_without_a_gpu(built):
    """No CPU fallback: on a box without a HIP device the engine refuses to exist."""
    from quakemigrate_amd.core import lib

    if lib.qmlib.qm_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(lib.QMHipError):
        lib.Engine(0)
    g = load_golden("ties_floor")
    with pytest.raises(lib.QMHipError):
        lib.migrate_and_find_max(g["onsets"], g["traveltimes"], int(g["fsmp"]),
                                 int(g["lsmp"]), int(g["available"]))


def test_shard_columns_boxes_cover_the_flat_range():
    """shard_columns / column_boxes: every rank's boxes tile its contiguous column range in
    ascending flat order; ranks are balanced to one column; partial boxes are one plane thick."""
    from quakemigrate_amd.distributed import column_boxes, shard_columns

    for nx, ny, world in [(201, 201, 8), (401, 401, 8), (5, 3, 4), (2, 2, 8), (17, 9, 2), (1, 1, 3)]:
        pos, sizes = 0, []
        for r in range(world):
            c0, c1 = shard_columns(nx, ny, world, r)
            assert c0 == pos
            boxes = column_boxes(c0, c1, ny)
            assert len(boxes) <= 3
            at = c0
            for x0, x1, y0, y1 in boxes:
                assert x0 * ny + y0 == at and x1 > x0 and y1 > y0
                assert (y0, y1) == (0, ny) or x1 == x0 + 1
                at += (x1 - x0) * (y1 - y0)
            assert at == c1
            pos = c1
            sizes.append(c1 - c0)
        assert pos == nx * ny and max(sizes) - min(sizes) <= 1


def test_shard_planes_partition():
    from quakemigrate_amd.distributed import shard_planes

    for nx, world in [(201, 8), (7, 8), (401, 3), (5, 1), (1, 2)]:
        spans = [shard_planes(nx, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == nx
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_neutral_partial_of_an_empty_slab_leaves_the_fold_unchanged():
    """More ranks than x-planes: ShardedDetector lets such a rank contribute (-inf, no index, 0);
    the fold over the ranks' sets must not see it."""
    import torch

    from quakemigrate_amd import distributed as qd

    rng = np.random.default_rng(5)
    ns, n_total = 64, 1000
    g = torch.empty((3, 3, ns), dtype=torch.float64)
    for r in range(2):
        g[r, 0] = torch.from_numpy(rng.integers(-8, 9, size=ns) / 4.0)      # many ties
        g[r, 1].view(torch.int64).copy_(torch.from_numpy(rng.integers(0, n_total, size=ns)))
        g[r, 2] = torch.from_numpy(rng.uniform(1.0, 2.0, size=ns))
    g[2, 0] = float("-inf")
    g[2, 1].view(torch.int64).fill_(qd.INT64_MAX)
    g[2, 2] = 0.0
    with_empty = qd.combine_packed_torch(g, n_total)
    without = qd.combine_packed_torch(g[:2].contiguous(), n_total)
    assert all(torch.equal(u, v) for u, v in zip(with_empty, without))
    assert int(with_empty[2].max()) < n_total


def test_scan_glue_shapes_and_errors(built):
    """MigrationScan validates before touching the GPU (no device needed here)."""
    from quakemigrate_amd import scan

    assert scan.time2sample(1.6, 50) == 80 and scan.time2sample(0.65, 250) == 162

    class Lut:
        def serve_traveltimes(self, sr, availability):
            raise KeyError("P")

    class Onset:
        def calculate_onsets(self, data):
            class OD:
                sampling_rate = 50
                availability = {"A_P": 1, "B_P": 0}
            return np.ones((1, 300)), OD()

    class NoEngine:                                    # (nothing of it is reached past the table)
        table_generation = 0

        def select_table(self, key, capacity=4):
            return False

    s = scan.MigrationScan(Lut(), Onset(), 1.0, 2.0, engine=NoEngine())
    with pytest.raises(scan.LUTPhasesException, match="phases"):
        s._compute(object())

    # parked tables are keyed by a process-unique name of the LUT object, not id() (a later object
    # at a dead LUT's address must not bring its table back): one per object, shared by the scans
    # over the same object; an object that takes no attributes gets a fresh one per scan
    lut_a, lut_b = Lut(), Lut()
    ta = scan.MigrationScan(lut_a, Onset(), 1.0, 2.0, engine=NoEngine())._lut_token
    assert scan.MigrationScan(lut_a, Onset(), 1.0, 2.0, engine=NoEngine())._lut_token == ta
    assert scan.MigrationScan(lut_b, Onset(), 1.0, 2.0, engine=NoEngine())._lut_token != ta

    class Slotted:
        __slots__ = ()
    frozen = Slotted()
    assert (scan.MigrationScan(frozen, Onset(), 1.0, 2.0, engine=NoEngine())._lut_token
            != scan.MigrationScan(frozen, Onset(), 1.0, 2.0, engine=NoEngine())._lut_token)


# ---------------------------------------------------------------------------------
# world_size-2 exchange on CPU (gloo): per-shard partials come from the oracle (the
# checker generating inputs), the exchange under test is quakemigrate_amd.distributed
# ---------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, str(ROOT))
    from oracle import qm_oracle
    from quakemigrate_amd import distributed as qd
    from quakemigrate_amd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grid = (17, 9, 8)
    x0, x1 = qd.shard_planes(grid[0], world, rank)
    case = synth.make_case("C2", step=5, grid=grid, rows=6, n_samples=211,
                           x_range=(x0, x1))
    # this rank's slab through the oracle: log-domain max, global argmax, sum
    vol = qm_oracle.c_migrate(case.onsets, case.traveltimes, case.fsmp, case.lsmp,
                              case.available, threads=2)
    vol = vol.reshape(-1, vol.shape[-1])
    local_idx = np.argmax(vol, axis=0)
    pmax = torch.from_numpy(np.log2(vol[local_idx, np.arange(vol.shape[1])]))
    pidx = torch.from_numpy(local_idx.astype(np.int64) + x0 * grid[1] * grid[2])
    psum = torch.from_numpy(vol.sum(axis=0))
    a, b, c = qd.exchange_partials(pmax, pidx, psum, int(np.prod(grid)))
    # the packed form ShardedDetector runs: one all-gather of [3][ns], then the fold
    packed = torch.empty((3, vol.shape[1]), dtype=torch.float64)
    packed[0], packed[2] = pmax, psum
    packed[1].view(torch.int64).copy_(pidx)
    gathered = torch.empty((world, 3, vol.shape[1]), dtype=torch.float64)
    qd.all_gather_packed(packed, gathered)
    pa, pb, pc = qd.combine_packed_torch(gathered, int(np.prod(grid)))
    assert torch.equal(pc, c) and torch.equal(pa, a)
    assert torch.allclose(pb, b, rtol=1e-14, atol=0)
    # the column partition (flat-index ranges, up to three boxes per rank): every box's partial
    # from the oracle, one all-gather of [3 boxes][3][ns], fold over world * 3 sets
    full = synth.make_case("C2", step=5, grid=grid, rows=6, n_samples=211)
    boxes = qd.column_boxes(*qd.shard_columns(grid[0], grid[1], world, rank), grid[1])
    assert 1 <= len(boxes) <= 3 and sum((b[1] - b[0]) * (b[3] - b[2]) for b in boxes) == \
        qd.shard_columns(grid[0], grid[1], world, rank)[1] - qd.shard_columns(grid[0], grid[1], world, rank)[0]
    cpacked = torch.empty((qd.MAX_BOXES, 3, vol.shape[1]), dtype=torch.float64)
    cpacked[:, 0] = float("-inf")
    cpacked[:, 1].view(torch.int64).fill_(qd.INT64_MAX)
    cpacked[:, 2] = 0.0
    for k, (bx0, bx1, by0, by1) in enumerate(boxes):
        box_tt = np.ascontiguousarray(full.traveltimes[bx0:bx1, by0:by1])
        bvol = qm_oracle.c_migrate(full.onsets, box_tt, full.fsmp, full.lsmp, full.available,
                                   threads=2)
        bvol = bvol.reshape(-1, bvol.shape[-1])
        bi = np.argmax(bvol, axis=0)
        cpacked[k, 0] = torch.from_numpy(np.log2(bvol[bi, np.arange(bvol.shape[1])]))
        cpacked[k, 1].view(torch.int64).copy_(
            torch.from_numpy(bi.astype(np.int64) + (bx0 * grid[1] + by0) * grid[2]))
        cpacked[k, 2] = torch.from_numpy(bvol.sum(axis=0))
    cgathered = torch.empty((world, qd.MAX_BOXES, 3, vol.shape[1]), dtype=torch.float64)
    qd.all_gather_packed(cpacked, cgathered)
    ca, cb, cc = qd.combine_packed_torch(cgathered.view(world * qd.MAX_BOXES, 3, -1),
                                         int(np.prod(grid)))
    assert torch.equal(cc, c) and torch.equal(ca, a)
    assert torch.allclose(cb, b, rtol=1e-14, atol=0)
    # the marginalised map of a locate window, slab by slab, gathered on every rank
    marg = torch.from_numpy(vol[:, 40:150].sum(axis=1).reshape(x1 - x0, grid[1], grid[2]))
    whole = qd.gather_planes(marg, grid[0])
    np.savez(pathlib.Path(tmp) / f"rank{rank}.npz", a=a.numpy(), b=b.numpy(), c=c.numpy(),
             marg=whole.numpy())
    dist.destroy_process_group()


def test_exchange_partials_world_size_2_gloo(oracle, tmp_path):
    import torch.multiprocessing as mp

    from quakemigrate_amd import synth

    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    grid = (17, 9, 8)
    case = synth.make_case("C2", step=5, grid=grid, rows=6, n_samples=211)
    want = oracle.detect(case.onsets, case.traveltimes, case.fsmp, case.lsmp,
                         case.available, threads=2)
    vol = oracle.c_migrate(case.onsets, case.traveltimes, case.fsmp, case.lsmp, case.available,
                           threads=2)
    for rank in range(2):
        got = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(got["c"], want[2])
        np.testing.assert_allclose(got["a"], want[0], rtol=1e-12)
        np.testing.assert_allclose(got["b"], want[1], rtol=1e-12)
        np.testing.assert_allclose(got["marg"], vol[..., 40:150].sum(axis=-1), rtol=1e-13)


def _worker_ties(rank, world, port, tmp):
    """tie_rule = 1 across ranks, stated with numpy / torch on CPU: the protocol of ShardedDetector's second
    exchange (csrc/qm_ties.hpp: tie_export_kernel / tie_fold_kernel) on the near_ties_bricks family, the grid cut
    at its mirror plane."""
    import ctypes

    import torch
    import torch.distributed as dist

    sys.path.insert(0, str(ROOT))
    from conftest import load_golden
    from quakemigrate_amd import distributed as qd
    from quakemigrate_amd.core import lib as qlib

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("near_ties_bricks")
    tt, fsmp, lsmp, avail = g["traveltimes"], int(g["fsmp"]), int(g["lsmp"]), int(g["available"])
    nx, ny, nz, S = tt.shape
    x0, x1 = qd.shard_planes(nx, world, rank)
    lon = np.log(np.clip(g["onsets"], 0.01, np.inf))
    ns = lon.shape[1] - fsmp - lsmp
    flat = np.clip(tt[x0:x1].reshape(-1, S), 0, None)
    stack = np.zeros((flat.shape[0], ns))
    for r in range(S):                                   # ascending rows, one add each (migratelib.c:54-59)
        stack += lon[r][fsmp + flat[:, r][:, None] + np.arange(ns)[None, :]]
    offset = x0 * ny * nz
    z = stack * (1.4426950408889634074 / avail)
    first = np.argmax(z, axis=0)
    packed = torch.empty((3, ns), dtype=torch.float64)
    packed[0] = torch.from_numpy(z[first, np.arange(ns)])
    packed[1].view(torch.int64).copy_(torch.from_numpy(first.astype(np.int64) + offset))
    packed[2] = torch.from_numpy(np.exp2(z).sum(axis=0))
    gathered = torch.empty((world, 3, ns), dtype=torch.float64)
    qd.all_gather_packed(packed, gathered)
    _, _, idx = qd.combine_packed_torch(gathered, nx * ny * nz)
    # this rank's candidates against the GRID's maxima: within the slack, compared on a correctly rounded exp
    zbest = gathered[:, 0, :].max(dim=0).values.numpy()
    lo = zbest - (4.0e-16 + 2.0 ** -50 * np.abs(zbest))
    tie = torch.zeros((2, ns), dtype=torch.float64)
    keys, at = tie[0].view(torch.int64), tie[1].view(torch.int64)
    at.fill_(qd.INT64_MAX)
    exp_cr = qlib.qmlib.qm_exp_correctly_rounded
    for t in range(ns):
        cand = np.nonzero(z[:, t] >= lo[t])[0]
        if len(cand) == 0:
            continue
        e = np.array([exp_cr(float(stack[n, t] * (1.0 / avail))) for n in cand])
        keys[t] = int(np.float64(e.max()).view(np.int64))
        at[t] = int(cand[np.argmax(e)]) + offset         # (first maximum = lowest index)
    tie_gathered = torch.empty((world, 2, ns), dtype=torch.float64)
    qd.all_gather_packed(tie, tie_gathered)
    refined = qd.fold_ties_torch(tie_gathered.view(torch.int64), idx)
    np.savez(pathlib.Path(tmp) / f"ties{rank}.npz", default=idx.numpy(), refined=refined.numpy(),
             candidates=int((keys != 0).sum()))

    # a detector whose ranks disagree on tie_rule says so on every rank at construction (it would hang in the
    # second exchange otherwise); ranks that agree construct
    class _NoTable:
        n_rows = None

        def __init__(self, rule):
            self.rule = rule

        def get(self, key):
            assert key == "tie_rule"
            return self.rule

    try:
        qd.ShardedDetector(_NoTable(rank), 10, 5, "cpu")
        raised = False
    except ValueError as e:
        raised = "tie_rule" in str(e)
    assert raised
    assert qd.ShardedDetector(_NoTable(1), 10, 5, "cpu").tie_rule is True
    assert qd.ColumnShardedDetector([], 10, 5, "cpu", fold_engine=_NoTable(0)).tie_rule is False
    dist.destroy_process_group()


def test_tie_rule_exchange_world_size_2_gloo(built, tmp_path):
    """The sharded form of tie_rule = 1 (SURVEY.md section 8e + migratelib.c:98-105): ranks exchange (largest
    correctly rounded exp among their near-tied nodes, lowest global index reaching it) behind the partials and
    fold -- on the mirror-twin family whose twins live on DIFFERENT ranks the folded index series equals the
    reference's scalar-libm build on every sample, where the default rule differs on several per cent."""
    import torch
    import torch.multiprocessing as mp

    from conftest import load_golden
    from quakemigrate_amd import distributed as qd

    mp.spawn(_worker_ties, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = load_golden("near_ties_bricks")["idx_scalar"]
    for rank in range(2):
        got = np.load(tmp_path / f"ties{rank}.npz")
        assert np.array_equal(got["refined"], want), float(np.mean(got["refined"] != want))
        assert 0.03 < np.mean(got["default"] != want) < 0.2
        assert got["candidates"] > 0.3 * len(want)           # (both ranks hold candidates for most samples)
    # the fold itself: a rank that followed too many candidate sets (-1) keeps the default on every rank; equal
    # exps go to the lowest index; samples nobody refined keep theirs
    keys = torch.tensor([[5, 7, 0, -1, 9], [5, 3, 0, 8, 9]], dtype=torch.int64)
    at = torch.tensor([[40, 11, qd.INT64_MAX, 3, 70], [30, 2, qd.INT64_MAX, 4, 60]], dtype=torch.int64)
    default = torch.tensor([100, 101, 102, 103, 104], dtype=torch.int64)
    got = qd.fold_ties_torch(torch.stack([keys, at], dim=1), default)
    assert got.tolist() == [30, 11, 102, 103, 60]


def _worker_columns(rank, world, port, tmp, grid, rows, ns, tag):
    """One rank of a column-partitioned detect on CPU: every box's partial set from the oracle, the packed
    all-gather + fold and the three-all-reduce form (quakemigrate_amd.distributed) -- SURVEY section 8e."""
    import torch
    import torch.distributed as dist

    sys.path.insert(0, str(ROOT))
    from oracle import qm_oracle
    from quakemigrate_amd import distributed as qd
    from quakemigrate_amd import synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = synth.make_case("C3", step=4, grid=grid, rows=rows, n_samples=ns)
    n_total = int(np.prod(grid))
    c0, c1 = qd.shard_columns(grid[0], grid[1], world, rank)
    boxes = qd.column_boxes(c0, c1, grid[1])
    assert len(boxes) <= qd.MAX_BOXES
    assert sum((b[1] - b[0]) * (b[3] - b[2]) for b in boxes) == c1 - c0
    packed = torch.empty((qd.MAX_BOXES, 3, ns), dtype=torch.float64)
    packed[:, 0] = float("-inf")
    packed[:, 1].view(torch.int64).fill_(qd.INT64_MAX)
    packed[:, 2] = 0.0
    for k, (bx0, bx1, by0, by1) in enumerate(boxes):
        box_tt = np.ascontiguousarray(full.traveltimes[bx0:bx1, by0:by1])
        bvol = qm_oracle.c_migrate(full.onsets, box_tt, full.fsmp, full.lsmp, full.available, threads=1)
        bvol = bvol.reshape(-1, ns)
        bi = np.argmax(bvol, axis=0)
        # local flat index inside the box -> flat index of the full grid
        lx, rem = np.divmod(bi, (by1 - by0) * grid[2])
        ly, lz = np.divmod(rem, grid[2])
        gidx = ((bx0 + lx) * grid[1] + (by0 + ly)) * grid[2] + lz
        packed[k, 0] = torch.from_numpy(np.log2(bvol[bi, np.arange(ns)]))
        packed[k, 1].view(torch.int64).copy_(torch.from_numpy(gidx.astype(np.int64)))
        packed[k, 2] = torch.from_numpy(bvol.sum(axis=0))
    gathered = torch.empty((world, qd.MAX_BOXES, 3, ns), dtype=torch.float64)
    qd.all_gather_packed(packed, gathered)
    a, b, c = qd.combine_packed_torch(gathered.view(world * qd.MAX_BOXES, 3, -1), n_total)
    # the all-reduce form on the rank's own fold of its boxes (a rank without columns brings the neutral set)
    own = packed.view(1, qd.MAX_BOXES, 3, ns)
    pm, order = own[0, :, 0].max(dim=0)
    ties = own[0, :, 0] == pm
    pi = torch.where(ties, own[0, :, 1].view(torch.int64), torch.full((1,), qd.INT64_MAX)).min(dim=0).values
    ps = own[0, :, 2].sum(dim=0)
    ra, rb, rc = qd.exchange_partials(pm.contiguous(), pi.contiguous(), ps.contiguous(), n_total)
    assert torch.equal(rc, c) and torch.equal(ra, a)
    assert torch.allclose(rb, b, rtol=1e-14, atol=0)
    np.savez(pathlib.Path(tmp) / f"{tag}{rank}.npz", a=a.numpy(), b=b.numpy(), c=c.numpy(),
             boxes=np.array([len(boxes), c1 - c0]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,grid,tag", [(8, (201, 201, 2), "c3columns"), (9, (2, 4, 3), "degenerate")])
def test_column_partition_fold_at_world_8_gloo(oracle, tmp_path, world, grid, tag):
    """The first 8-GPU run must not be the first time the world-8 arithmetic executes:
    C3's real column partition -- 201 x 201 columns over 8 ranks = 5051 / 5050 each, one to three boxes per
    rank -- with every box's partial from the oracle, folded by the packed all-gather and by the all-reduce
    form, against the oracle's one-shot series (index and maximum exact, normalised value to 1e-14); and nine
    ranks on eight columns: one rank holds nothing."""
    import torch.multiprocessing as mp

    from quakemigrate_amd import distributed as qd
    from quakemigrate_amd import synth

    rows, ns = 3, 48
    port = _free_port()
    mp.spawn(_worker_columns, args=(world, port, str(tmp_path), grid, rows, ns, tag), nprocs=world, join=True)
    case = synth.make_case("C3", step=4, grid=grid, rows=rows, n_samples=ns)
    vol = oracle.c_migrate(case.onsets, case.traveltimes, case.fsmp, case.lsmp, case.available, threads=4)
    vol = vol.reshape(-1, ns)
    idx = np.argmax(vol, axis=0)
    top = vol[idx, np.arange(ns)]
    counts = []
    for rank in range(world):
        got = np.load(tmp_path / f"{tag}{rank}.npz")
        assert np.array_equal(got["c"], idx), rank
        np.testing.assert_allclose(got["a"], top, rtol=4e-16)                 # (2^log2 of the exact maximum)
        # (against the sum over the nodes in extended precision: float64 sums of 8e4 terms in two different
        # orders differ by 3e-14 from each other)
        exact = np.asarray(vol.astype(np.longdouble).sum(axis=0), dtype=np.float64)
        np.testing.assert_allclose(got["b"], top * vol.shape[0] / exact, rtol=1e-14)
        counts.append(tuple(int(v) for v in got["boxes"]))
    columns = [c for _, c in counts]
    assert sum(columns) == grid[0] * grid[1] and max(columns) - min(columns) <= 1
    if tag == "c3columns":
        assert sorted(set(columns)) == [5050, 5051] and all(1 <= n <= 3 for n, _ in counts)
        assert {n for n, _ in counts} >= {2, 3}                              # partitions with several boxes occur
    else:
        assert columns.count(0) == 1 and [n for n, c in counts if c == 0] == [0]


def test_cubic_rbf_algebra_is_scipys_rbf():
    """locate._cubic_rbf_weights / _cubic_rbf_on_grid (what Engine.rbf_peak evaluates on the GPU)
    against scipy.interpolate.Rbf(function="cubic") called the way _splineloc calls it
    (scan.py:777-804: default "xy" meshgrids of the window and of the 10x finer grid)."""
    from scipy.interpolate import Rbf

    from quakemigrate_amd import locate

    rng = np.random.default_rng(8)
    n, upscale = 5, 10
    g = np.indices((n, n, n)).astype(np.float64)
    sub = np.exp(-((g[0] - 2.3) ** 2 + (g[1] - 1.6) ** 2 + (g[2] - 2.1) ** 2) / 2.5)
    sub += 0.01 * rng.random(sub.shape)
    c = np.arange(n, dtype=np.float64)
    x, y, z = np.meshgrid(c, c, c)
    rbf = Rbf(x.ravel(), y.ravel(), z.ravel(), sub.ravel(), function="cubic")
    f = np.linspace(0, n - 1, (n - 1) * upscale + 1)
    xf, yf, zf = np.meshgrid(f, f, f)
    want = rbf(xf.ravel(), yf.ravel(), zf.ravel()).reshape(xf.shape)
    got = locate._cubic_rbf_on_grid(sub, upscale)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12)
    assert np.unravel_index(np.argmax(got), got.shape) == np.unravel_index(np.argmax(want), want.shape)
    np.testing.assert_allclose(locate._cubic_rbf_weights(sub).ravel(), rbf.nodes, rtol=1e-8,
                               atol=1e-12)


def _cube(a, centre, width):
    out = np.full((width,) * 3, np.nan)
    half = width // 2
    for i in range(width):
        for j in range(width):
            for k in range(width):
                x, y, z = centre[0] - half + i, centre[1] - half + j, centre[2] - half + k
                if 0 <= x < a.shape[0] and 0 <= y < a.shape[1] and 0 <= z < a.shape[2]:
                    out[i, j, k] = a[x, y, z]
    return out


@pytest.mark.parametrize("name", ["corner", "interior_even", "interior_odd", "near_face", "thin"])
def test_location_window_algebra_matches_reference(name):
    """The host half of locate.calculate_location (Gaussian / spline fits on the windows the
    engine returns) against the reference's _gaufit3d / _splineloc outputs."""
    from conftest import load_golden
    from quakemigrate_amd import locate

    g = load_golden("locate_fits")
    spacing = g[f"{name}_node_spacing"]
    coa, smoothed = g[f"{name}_coa_map"], g[f"{name}_smoothed"]
    peak = np.array(np.unravel_index(np.nanargmax(coa), coa.shape))
    speak = np.array(np.unravel_index(np.nanargmax(smoothed), smoothed.shape))
    loc, sigma, _ = locate.gaussian_from_window(_cube(smoothed, speak, 7), np.nanmean(smoothed),
                                                speak, smoothed.shape)
    np.testing.assert_allclose(loc, g[f"{name}_gaussian"], rtol=1e-9)
    np.testing.assert_allclose(sigma * spacing, g[f"{name}_gaussian_uncertainty"], rtol=1e-9)
    got = locate.spline_from_window(_cube(coa, peak, 5), peak, coa.shape)
    assert np.array_equal(got, g[f"{name}_spline"])


def _screen_scale(L, S, available):
    """k and the quantised onsets exactly as screen_quantise_kernel computes them."""
    c = 1.4426950408889634 / available
    rmax = float(np.abs(L).max())
    k = 30
    while k > 0 and S * (rmax * c * 2.0 ** k + 1.0) >= 2147483647.0:
        k -= 1
    q = np.rint(L * (c * 2.0 ** k)).astype(np.int64)
    return k, c, q


def test_fixed_point_screening_bounds_hold_on_cpu():
    """The two claims the screened detect rests on (qm_screen.hpp), restated in NumPy on
    adversarial scales: (1) with q = rint(L c 2^k), integer stacks Q are within S/2 units of the
    float64 z 2^k, so the float64 arg-max node has Q >= Qmax - (S + 2): it is always among the
    candidates; (2) the terms exp2f(float32(Q) 2^-k) of 32 nodes, summed in float32 with Kahan
    compensation, are within 6.7e-7 relative of the float64 sum whenever the preconditions hold
    -- also for CORRELATED inputs (constant rows: every node stacks the same values)."""
    rng = np.random.default_rng(31)
    worst = 0.0
    for trial in range(60):
        S = int(rng.integers(2, 65))
        available = S if trial % 3 else max(1, S // 2)
        n_nodes, T = 800, 96
        if trial % 4 == 0:                                           # constant rows: correlated
            L = np.log(rng.uniform(0.4, 9.0, size=(S, 1))) * np.ones((1, T))
        else:
            scale = 10.0 ** rng.uniform(-1.5, 1.5)
            L = np.log(np.clip(rng.lognormal(0, 1.0, size=(S, T)) * scale, 0.01, np.inf))
        k, c, q = _screen_scale(L, S, available)
        dz = S * 2.0 ** -(k + 1)
        if not (np.log(2) * dz <= 1.0e-7 and np.abs(L).max(axis=1).sum() * c <= 8.0):
            continue                                                  # the device redoes such a step
        assert np.abs(q).max() * S < 2 ** 31
        tt = rng.integers(0, T, size=(n_nodes, S))
        rows = np.arange(S)[None, :]
        z64 = np.zeros(n_nodes)
        for r in range(S):
            z64 += L[r, tt[:, r]]                                     # ascending rows, float64
        z64 *= c
        Q = q[rows, tt].sum(axis=1)
        assert np.abs(Q * 2.0 ** -k - z64).max() <= dz * (1 + 1e-9)   # (1) the stack bound
        assert Q[int(np.argmax(z64))] >= Q.max() - (S + 2)            #     the candidate rule
        z32 = Q.astype(np.float32) * np.float32(2.0 ** -k)            # (2) one rounding
        term = np.exp2(z32).astype(np.float32)                        #     <= 1 ulp
        part = term.reshape(-1, 32)                                   #     a wave's nodes of a brick
        acc = np.zeros(part.shape[0], dtype=np.float32)
        comp = np.zeros_like(acc)
        for i in range(32):                                           #     compensated float32 sum
            y = part[:, i] - comp
            t = acc + y
            comp = (t - acc) - y
            acc = t
        want = np.exp2(z64).reshape(-1, 32).sum(axis=1)
        rel = np.abs(acc.astype(np.float64) - want) / want
        worst = max(worst, float(rel.max()))
        assert rel.max() <= 6.7e-7
    assert worst > 0.0




def test_generated_shift_loop_is_current_whatever_the_environment_holds():
    """qm_shift_asm.inc is generated (gen_shift_asm.py) and committed: the two must agree -- and the product
    generator reads NOTHING from the environment (a stray QM_SHIFT_EXP=noreads used to
    yield a library that built, passed this test and was wrong).  Variants come from tools/dev/shift_overlay.py
    only, and say so in the file they write."""
    import subprocess
    import sys

    csrc = ROOT / "quakemigrate_amd" / "csrc"
    hostile = dict(os.environ, QM_SHIFT_EXP="noreads,nowait", QM_SHIFT_NQMIN="6", QM_SHIFT_PF="0",
                   QM_SHIFT_PACKED="0", QM_SHIFT_STAGE_IN_LOOP="0", QM_SHIFT_VB="40", QM_DEV_VARIANT="1")
    gen = subprocess.check_output([sys.executable, str(csrc / "gen_shift_asm.py")], text=True, env=hostile)
    assert gen == (csrc / "qm_shift_asm.inc").read_text()
    assert "os.environ" not in (csrc / "gen_shift_asm.py").read_text()
    variant = subprocess.check_output([sys.executable, str(ROOT / "tools" / "dev" / "shift_overlay.py"),
                                       "PF_AHEAD=32"], text=True)
    assert variant != gen and "overlay=PF_AHEAD=32" in variant and "overlay=none" in gen


def test_stored_hbm_traffic_was_measured_on_this_trees_kernels():
    """bench.py reports `roofline.traffic` from the newest profiles/r*_pmc_traffic.json only while the stacking
    kernels' sources still hash to what the PMC passes ran on (tools/pmc_traffic.py: kernel_code_digest) -- a
    stale file turns the driver's bench line's traffic into null.  The committed file must belong to the
    committed kernels (a comment edit in qm_shift.hpp counts: re-run `tools/round_evidence.sh <tag> pmc`)."""
    import glob
    import json

    sys.path.insert(0, str(ROOT / "tools"))
    try:
        import pmc_traffic
    finally:
        sys.path.pop(0)
    files = sorted(glob.glob(str(ROOT / "profiles" / "r*_pmc_traffic.json")))
    assert files, "no stored traffic figure"
    stored = json.load(open(files[-1]))
    assert stored["_kernel_code"] == pmc_traffic.kernel_code_digest(), files[-1]
    assert stored["C3:detect"]["kernel"] == "void qm::stack_shift_kernel<0, 8>"      # (the name bench.py asks for)


def test_library_exports_only_the_c_abi_and_says_what_it_was_built_from(built):
    """`nm -D`: the symbols include/qmhip.h declares and nothing else (csrc/qmhip.map; the kernels' host stubs
    and the engine's C++ internals used to be visible beside them); qm_build_info() names the product
    generator's constants, the digest of its source, no overlay, no development defines."""
    import hashlib
    import subprocess

    csrc = ROOT / "quakemigrate_amd" / "csrc"
    out = subprocess.check_output(["nm", "-D", "--defined-only", str(csrc / "libqmhip.so")], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == _declared_functions(), sorted(set(exported) ^ set(_declared_functions()))
    lib = ctypes.CDLL(str(csrc / "libqmhip.so"))
    lib.qm_build_info.restype = ctypes.c_char_p
    info = lib.qm_build_info().decode()
    digest = hashlib.sha256((csrc / "gen_shift_asm.py").read_bytes()).hexdigest()[:16]
    assert f"generator={digest}" in info and "overlay=none" in info and info.endswith("defines=none"), info
    assert "NQMAX=6 NQMIN=4" in info and "PF_AHEAD=16" in info


def test_headline_kernels_stay_in_registers(tmp_path):
    """Compile the kernels the BASELINE configs run (gfx950, no GPU needed) and read the
    compiler's resource summary: no scratch (a spill inside the node loop costs 20-30x, as two
    versions of this code found out), at most 128 VGPRs (4 wavefronts per SIMD, which the
    workgroup layouts are sized for)."""
    import re
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not pathlib.Path(hipcc).exists():
        pytest.skip("no hipcc")
    src = tmp_path / "k.hip"
    src.write_text(
        '#include <hip/hip_runtime.h>\n#include "qm_kernels.hpp"\n#include "qm_pair.hpp"\n'
        "template __global__ void qm::stack_exact_kernel<4, false, 30>(qm::StackArgs);\n"   # C3
        "template __global__ void qm::stack_exact_kernel<4, false, 20>(qm::StackArgs);\n"   # C2
        "template __global__ void qm::stack_exact_kernel<4, false, 60>(qm::StackArgs);\n"   # C4
        "template __global__ void qm::stack_exact_kernel<2, false, 60>(qm::StackArgs);\n"
        "template __global__ void qm::stack_pair_kernel<2, true, 30>(qm::StackArgs);\n"     # locate
        "template __global__ void qm::stack_exact_marginal_kernel<4, 30>(qm::StackArgs);\n"
        "template __global__ void qm::stack_lds_kernel<2, false, 3>(qm::StackArgs);\n")     # C1
    # the shift-reuse detect kernel (C3 / C5 since round 3): two wavefronts per SIMD by design (64
    # accumulators + two register windows of 24 doubles), i.e. at most 256 VGPRs, no scratch
    shift = tmp_path / "s.hip"
    shift.write_text('#define QM_SHIFT_TU 1\n#include <hip/hip_runtime.h>\n#include "qm_shift.hpp"\n'
                     "template __global__ void qm::stack_shift_kernel<0, 4>(qm::ShiftArgs);\n"
                     "template __global__ void qm::stack_shift_kernel<1, 4>(qm::ShiftArgs);\n"
                     "template __global__ void qm::stack_shift_kernel<2, 4>(qm::ShiftArgs);\n"
                     "template __global__ void qm::stack_shift_kernel<0, 8>(qm::ShiftArgs);\n"
                     "template __global__ void qm::stack_shift_kernel<1, 8>(qm::ShiftArgs);\n"
                     "template __global__ void qm::stack_shift_kernel<2, 8>(qm::ShiftArgs);\n"
                     "template __global__ void qm::stack_shift_kernel<0, 12>(qm::ShiftArgs);\n"
                     "template __global__ void qm::stack_shift_rows_kernel<8>(qm::ShiftArgs);\n"
                     "template __global__ void qm::stack_shift_rows2_kernel<false, 8>(qm::ShiftArgs);\n"
                     "template __global__ void qm::stack_shift_rows2_kernel<true, 8>(qm::ShiftArgs);\n"
                     "template __global__ void qm::stack_shift_rows4_kernel<false>(qm::ShiftArgs);\n"
                     "template __global__ void qm::stack_shift_rows4_kernel<true>(qm::ShiftArgs);\n"
                     # (round 6: the fused detect that also leaves a row of maxima per brick, tie_rule = 1)
                     "template __global__ void qm::stack_shift_bricks_kernel<4>(qm::ShiftArgs);\n"
                     "template __global__ void qm::stack_shift_bricks_kernel<8>(qm::ShiftArgs);\n")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17",
                           f"-I{ROOT / 'quakemigrate_amd' / 'csrc'}", "-c", str(shift), "-o",
                           str(tmp_path / "s.o"), "--save-temps"], cwd=tmp_path,
                          stderr=subprocess.DEVNULL)
    sasm = next(tmp_path.glob("s-hip-amdgcn-*.s")).read_text()
    found = re.findall(r"\.set (\S*stack_shift\w*_kernel\S*)\.num_vgpr, (\d+)", sasm)
    assert len(found) == 15, found                          # (+ the wide tiles' row-block kernel: not a template)
    for name, vgprs in found:
        if "Li12E" in name:                                # the opt-in 12-wave shape: three per SIMD
            assert int(vgprs) <= 168, (name, vgprs)
            continue
        assert int(vgprs) <= 256, (name, vgprs)
        sscr = re.search(re.escape(name) + r"\.private_seg_size, (\d+)", sasm)
        assert sscr is not None and int(sscr.group(1)) <= 64, name
        # (the four inlined tile bodies of a kernel can leave a dead spill slot behind -- a few bytes
        # of private segment that nothing accesses; what must not exist is scratch TRAFFIC)
        body = sasm[sasm.index(name + ":"):]
        body = body[:body.index(".Lfunc_end")]
        assert "scratch_" not in body and "buffer_store" not in body and "buffer_load" not in body, name
    # The row-block kernel keeps a group's accumulators in the generated loop's hard registers
    # ACROSS inline-asm statements: the compiler's own code (staging, barriers) must never touch a
    # register from kShiftBlockVgprs up.  Its attributes make those reserved; check the ISA.
    # (the same walk runs inside every build of the unit: __graft_entry__.check_row_block_registers)
    sys.path.insert(0, str(ROOT / "quakemigrate_amd" / "csrc"))
    try:
        import check_shift_isa
    finally:
        sys.path.pop(0)
    inc_text = (ROOT / "quakemigrate_amd" / "csrc" / "qm_shift_asm.inc").read_text()
    first_hard = check_shift_isa.first_hard_register(inc_text)
    assert check_shift_isa.check(sasm, first_hard) > 300
    # (round 6: the wide tiles' row-block kernel keeps 96 accumulators from kShiftWideBlockVgprs up)
    wide_first = check_shift_isa.first_hard_register(inc_text, wide=True)
    assert check_shift_isa.check_all(sasm, inc_text) > 400
    at = sasm.index(";;#ASMSTART", sasm.index(check_shift_isa.WIDE_ROW_BLOCK_KERNELS[0] + ":"))
    bad = sasm[:at] + "\tv_mov_b32_e32 v%d, 0\n" % (wide_first + 95) + sasm[at:]
    with pytest.raises(AssertionError, match="compiler code touches"):
        check_shift_isa.check_all(bad, inc_text)
    # ... and it does catch a violation: a compiler-looking instruction on v80 outside the asm blocks
    at = sasm.index(";;#ASMSTART", sasm.index(check_shift_isa.ROW_BLOCK_KERNELS[0] + ":"))
    bad = sasm[:at] + "\tv_mov_b32_e32 v%d, 0\n" % first_hard + sasm[at:]
    with pytest.raises(AssertionError, match="compiler code touches"):
        check_shift_isa.check(bad, first_hard)
    assert sasm.count("global_load_lds_dwordx4") >= 12     # the second form stages straight into LDS
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17",
                           f"-I{ROOT / 'quakemigrate_amd' / 'csrc'}", "-c", str(src), "-o",
                           str(tmp_path / "k.o"), "--save-temps"], cwd=tmp_path,
                          stderr=subprocess.DEVNULL)
    asm = next(tmp_path.glob("k-hip-amdgcn-*.s")).read_text()
    seen = 0
    for name, vgprs in re.findall(r"\.set (\S*(?:stack_\w+_kernel)\S*)\.num_vgpr, (\d+)", asm):
        scratch = re.search(re.escape(name) + r"\.private_seg_size, (\d+)", asm)
        assert int(vgprs) <= 128, (name, vgprs)
        assert scratch is not None and int(scratch.group(1)) == 0, (name, scratch and scratch.group(1))
        seen += 1
    assert seen == 7


def _reference_binding_against(so_path):
    """The reference's own quakemigrate/core/lib.py, imported unmodified, with `_load_cdll`
    answering with `so_path` (the technique of oracle/make_golden.py: empty parent packages and
    a pass-through `util.timeit` stand in for imports that file never executes here)."""
    import importlib.util
    import sys
    import types

    ref = pathlib.Path("/root/reference/quakemigrate/core/lib.py")
    if not ref.exists():
        pytest.skip("the reference tree is not on this box")
    saved = {k: sys.modules.get(k) for k in ("quakemigrate", "quakemigrate.core",
                                             "quakemigrate.util", "quakemigrate.core.libnames",
                                             "quakemigrate.core.lib")}
    qm = types.ModuleType("quakemigrate"); qm.__path__ = []
    core = types.ModuleType("quakemigrate.core"); core.__path__ = []
    util = types.ModuleType("quakemigrate.util"); util.timeit = lambda *a, **k: (lambda f: f)
    ln = types.ModuleType("quakemigrate.core.libnames")
    ln._load_cdll = lambda name: ctypes.CDLL(str(so_path))
    qm.util = util
    sys.modules.update({"quakemigrate": qm, "quakemigrate.core": core, "quakemigrate.util": util,
                        "quakemigrate.core.libnames": ln})
    try:
        spec = importlib.util.spec_from_file_location("quakemigrate.core.lib", ref)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)                    # binds all five symbols (lib.py:24-283)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def test_reference_front_end_binds_the_drop_in_library(built):
    """copy qmlib<EXT_SUFFIX> over the reference's extension and its
    unmodified core/lib.py imports -- every `qmlib.<symbol>.argtypes = ...` line finds its symbol
    -- and the three STA/LTA entry points (host code in both libraries) give the reference's
    vectors through the reference's own wrappers."""
    import sysconfig

    alias = ROOT / "quakemigrate_amd" / "csrc" / ("qmlib" + sysconfig.get_config_var("EXT_SUFFIX"))
    ref_lib = _reference_binding_against(alias)
    for name in ("migrate", "find_max_coa", "overlapping_sta_lta", "centred_sta_lta",
                 "recursive_sta_lta"):
        assert getattr(ref_lib.qmlib, name).argtypes is not None, name
    g = load_golden("stalta")
    toy = g["toy"]
    assert (ref_lib.overlapping_sta_lta(toy, 2, 3)
            == np.array([1.0, 1.0, 1.5, 1.25, 21.0 / 18, 27.0 / 24])).all()      # test_onsets.py:27-35
    for kind in ("overlapping", "centred", "recursive"):
        fn = getattr(ref_lib, f"{kind}_sta_lta")
        np.testing.assert_allclose(fn(toy, 2, 3), g[f"toy_{kind}"], rtol=1e-15)
        np.testing.assert_allclose(fn(g["signal"], int(g["nsta"]), int(g["nlta"])), g[kind],
                                   rtol=1e-12)


def test_timeit_logs_the_references_line(caplog):
    """quakemigrate/util.py:651-669: one 'Elapsed time' line per call, info or debug level."""
    import logging

    from quakemigrate_amd.core import lib

    @lib.timeit("info")
    def loud():
        return 3

    @lib.timeit()
    def quiet():
        return 4

    with caplog.at_level(logging.DEBUG):
        assert loud() == 3 and quiet() == 4
    msgs = [(r.levelno, r.getMessage()) for r in caplog.records]
    assert [lvl for lvl, _ in msgs] == [logging.INFO, logging.DEBUG]
    assert all(m.startswith(" " * 21 + "Elapsed time: ") and m.endswith(" seconds.") for _, m in msgs)
    assert lib.migrate.__wrapped__ is not None and lib.find_max_coa.__name__ == "find_max_coa"


def test_exp_correctly_rounded_is_correctly_rounded(built):
    """qm_exp_correctly_rounded (csrc/qm_ties.hpp: the function the opt-in tie_rule = 1 compares
    near-tied nodes on; host + device code, the same source) against mpmath's exp rounded to
    nearest, over the range coalescence exponents live in and beyond -- every argument, every bit."""
    mpmath = pytest.importorskip("mpmath")
    from quakemigrate_amd.core import lib

    f = lib.qmlib.qm_exp_correctly_rounded
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.uniform(-12, 12, 3000), rng.uniform(0, 3, 3000), rng.uniform(-1e-3, 1e-3, 500),
                         rng.uniform(-700, 700, 500), [0.0, -0.0, 1.0, -1.0, 0.5, 709.0, -740.0, 1e-300]])
    with mpmath.workprec(200):
        want = np.array([float(mpmath.exp(mpmath.mpf(float(x)))) for x in xs])
    got = np.array([f(float(x)) for x in xs])
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:5]
    assert np.isnan(f(float("nan"))) and f(800.0) == np.inf and f(-800.0) == 0.0
