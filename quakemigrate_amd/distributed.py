# -*- coding: utf-8 -*-
"""
Multi-GPU sharding of the migrate / find_max_coa path.

Nodes are independent in ``migrate`` and ``find_max_coa`` is a reduction over
nodes, so the 3-D grid is sharded by contiguous ranges of x-planes (= contiguous
ranges of flat node indices, flat = (ix*ny+iy)*nz+iz, quakemigrate/lut/lut.py:165-166):
one process per GPU, each with its slab of the travel-time table resident, the
(small) onset array replicated.  Per timestep every rank produces its partial
``(log2-domain max, global argmax, sum of coalescence)`` per sample with
``Engine.detect_partial``; the path's only exchange is over ``n_samples`` elements
(48 KB per series at 6000 samples) -- RCCL over xGMI on the GPU box, gloo in the
CPU tests.  Two equivalent forms:

* **packed** (what :class:`ShardedDetector` runs): the three series live in ONE
  ``[3][n_samples]`` float64 buffer (the int64 indices as bit patterns), ONE
  all-gather moves every rank's buffer to every rank (``[world][3][n_samples]``),
  and the engine's ``combine_kernel`` (``Engine.finalize_packed``) folds the
  ``world`` sets on the device -- one collective launch and one kernel per step,
  no temporaries;
* **three all-reduces** (:func:`exchange_partials`, the plainly readable statement
  kept as the tested reference of the packed form):

      gmax = all_reduce(pmax, MAX)
      gidx = all_reduce(where(pmax == gmax, pidx, INT64_MAX), MIN)   # lowest index wins
      gsum = all_reduce(psum, SUM)
      max_coa = 2**gmax;  max_norm_coa = max_coa * n_nodes_total / gsum

Both reproduce the reference's tie-break (strict '>' in ascending node order,
migratelib.c:102) exactly, because equal maxima on two ranks resolve to the lower
global index.  Engines configured with ``tie_rule = 1`` (the reference's rule on NEAR-ties,
csrc/qm_ties.hpp) add a second, smaller exchange behind the first: every rank examines its
own partial sets against the grid's maxima (``Engine.tie_partial``), one all-gather of a
``[2][n_samples]`` buffer (largest correctly rounded exp, lowest global index reaching it)
and a device fold (``Engine.tie_fold``) give every rank the refined index series.  torch
supplies the process group and the device buffers -- plumbing, not the hot path.
"""

from __future__ import annotations

import torch

INT64_MAX = torch.iinfo(torch.int64).max


def shard_planes(nx: int, world_size: int, rank: int):
    """[x0, x1) of the x-planes owned by ``rank`` (balanced, contiguous)."""
    base, extra = divmod(nx, world_size)
    x0 = rank * base + min(rank, extra)
    return x0, x0 + base + (1 if rank < extra else 0)


def shard_columns(nx: int, ny: int, world_size: int, rank: int):
    """
    [c0, c1) of the (x, y) COLUMNS owned by ``rank``: the contiguous flat-index range
    ``[c0 * nz, c1 * nz)`` of SURVEY.md section 8(e), balanced to one column (nz nodes).  Plane
    slabs (:func:`shard_planes`) leave 201 planes over 8 ranks 26 / 25 -- 3.5 % of the step spent
    waiting for the ranks with 26; columns leave 5051 / 5050.
    """
    return shard_planes(nx * ny, world_size, rank)


def column_boxes(c0: int, c1: int, ny: int):
    """
    The column range [c0, c1) as at most three boxes ``(x0, x1, y0, y1)`` in ascending flat
    order: the rest of a first, partly owned x-plane, the whole planes, the start of a last,
    partly owned plane.  Each box is a contiguous flat range starting at ``(x0 * ny + y0) * nz``
    (a partial box is one plane thick), i.e. something an Engine can hold with that node offset.
    """
    boxes = []
    if c1 <= c0:
        return boxes
    xa, ya = divmod(c0, ny)
    xb, yb = divmod(c1, ny)
    if xa == xb:
        return [(xa, xa + 1, ya, yb)]
    if ya:
        boxes.append((xa, xa + 1, ya, ny))
        xa += 1
    if xb > xa:
        boxes.append((xa, xb, 0, ny))
    if yb:
        boxes.append((xb, xb + 1, 0, yb))
    return boxes


def combine_partials_local(pmax, pidx, psum, n_nodes_total):
    """
    Combine partial sets stacked along dim 0 ([n_sets, n_samples]) on one device;
    same arithmetic as the cross-rank exchange (used for tests and for combining
    several engines inside one process).
    """
    gmax = pmax.max(dim=0).values
    cand = torch.where(pmax == gmax.unsqueeze(0), pidx,
                       torch.full_like(pidx, INT64_MAX))
    gidx = cand.min(dim=0).values
    gsum = psum.sum(dim=0)
    peak = torch.exp2(gmax)
    return peak, peak * float(n_nodes_total) / gsum, gidx


def exchange_partials(pmax, pidx, psum, n_nodes_total, group=None):
    """
    Cross-rank combination of this rank's partial (three 1-D tensors of length
    n_samples) with three all-reduces.  Returns ``(max_coa, max_norm_coa,
    max_coa_idx)`` on every rank.
    """
    import torch.distributed as dist

    gmax = pmax.clone()
    dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
    gidx = torch.where(pmax == gmax, pidx, torch.full_like(pidx, INT64_MAX))
    dist.all_reduce(gidx, op=dist.ReduceOp.MIN, group=group)
    gsum = psum.clone()
    dist.all_reduce(gsum, op=dist.ReduceOp.SUM, group=group)
    peak = torch.exp2(gmax)
    return peak, peak * float(n_nodes_total) / gsum, gidx


def _backend(group=None):
    import torch.distributed as dist

    return str(dist.get_backend(group)).lower()


def all_gather_packed(packed, gathered, group=None):
    """
    ONE all-gather of this rank's packed partial ``[3][n_samples]`` into ``gathered``
    ``[world][3][n_samples]`` (rank order).  RCCL moves device buffers directly; gloo (the CPU
    tests, and the one-GPU plumbing runs) has no device all-gather, so device tensors are staged
    through the host there.
    """
    import torch.distributed as dist

    # flat views: the collective concatenates the ranks' buffers in rank order
    if packed.is_cuda and _backend(group) == "gloo":
        host = torch.empty(gathered.numel(), dtype=gathered.dtype)
        dist.all_gather_into_tensor(host, packed.reshape(-1).cpu(), group=group)
        gathered.view(-1).copy_(host)
    else:
        dist.all_gather_into_tensor(gathered.view(-1), packed.view(-1), group=group)
    return gathered


def _agreed_on_all_ranks(flag, device, group=None):
    """``flag`` (a bool every rank brings) if all ranks of the group bring the same one; ValueError otherwise.  One
    small all-reduce at construction: a detector whose ranks disagree on ``tie_rule`` would leave some of them
    waiting in the second exchange for the others."""
    import torch.distributed as dist

    on = torch.device("cpu") if _backend(group) == "gloo" else device
    both = torch.tensor([int(bool(flag)), -int(bool(flag))], dtype=torch.int64, device=on)
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)          # (max, -min) in one message
    if int(both[0]) != -int(both[1]):
        raise ValueError("the ranks of a sharded detector must share their engines' tie_rule "
                         f"(this rank: {int(bool(flag))})")
    return bool(flag)


def fold_ties_torch(tie_gathered, idx):
    """The fold of ``Engine.tie_fold`` stated with torch ops (CPU tests): ``tie_gathered`` int64 bit
    patterns ``[n_sets][2][n_samples]`` (exp keys -- positive doubles order as integers; -1 = a rank
    that followed too many candidate sets --, global indices), ``idx`` the default rule's series."""
    keys, at = tie_gathered[:, 0, :], tie_gathered[:, 1, :]
    overflow = (keys == -1).any(dim=0)
    best = keys.max(dim=0).values
    cand = torch.where(keys == best.unsqueeze(0), at, torch.full_like(at, INT64_MAX)).min(dim=0).values
    take = (best > 0) & ~overflow & (cand != INT64_MAX)
    return torch.where(take, cand, idx)


def combine_packed_torch(gathered, n_nodes_total):
    """The fold of ``Engine.finalize_packed`` stated with torch ops (CPU tests; any device)."""
    return combine_partials_local(gathered[:, 0, :], gathered[:, 1, :].view(torch.int64),
                                  gathered[:, 2, :], n_nodes_total)


def gather_planes(local, nx_total, group=None):
    """
    All-gather of x-plane slabs: ``local`` is this rank's ``(x1 - x0, ny, nz)`` piece of a map
    sharded with :func:`shard_planes`; returns the whole ``(nx_total, ny, nz)`` map on every rank.
    Used for the marginalised coalescence map of a locate window ("the
    marginal 3-D map is gathered (N doubles)").  Slabs differ by at most one plane, so every rank
    pads to the largest slab and the padding is cut away after the exchange.
    """
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = [shard_planes(nx_total, world, r) for r in range(world)]
    biggest = max(x1 - x0 for x0, x1 in sizes)
    padded = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype,
                         device=local.device)
    padded[: local.shape[0]] = local
    whole = torch.empty((world,) + tuple(padded.shape), dtype=local.dtype, device=local.device)
    if local.is_cuda and _backend(group) == "gloo":      # no device all-gather in gloo
        host = torch.empty(whole.numel(), dtype=whole.dtype)
        dist.all_gather_into_tensor(host, padded.reshape(-1).cpu(), group=group)
        whole.view(-1).copy_(host)
    else:
        dist.all_gather_into_tensor(whole.view(-1), padded.view(-1), group=group)
    return torch.cat([whole[r, : x1 - x0] for r, (x0, x1) in enumerate(sizes)], dim=0)


class ShardedDetector:
    """
    One rank's share of a grid-sharded detect sweep.

    ``engine`` has this rank's slab resident (``load_lut(slab, node_offset=...)``);
    ``detect(log_onsets_dev, fsmp, lsmp, available)`` returns the global series (device
    tensors, valid on torch's current stream).

    Stream discipline: the engine's kernels and the collective must be ordered on ONE stream.
    The detector therefore binds the engine to torch's current stream on ``device`` before every
    step (an engine left on its private stream would let the collective read the partials before
    the kernels have written them).

    ``exchange``: ``"packed"`` (default: one all-gather + the engine's device-side fold) or
    ``"allreduce"`` (three all-reduces, :func:`exchange_partials`).

    More ranks than x-planes: a rank whose slab is empty (``shard_planes`` gave ``x0 == x1``) passes
    an engine WITHOUT a resident table; it contributes the neutral partial (maximum ``-inf``, no
    index, sum 0) and takes part in every collective like the others.
    """

    def __init__(self, engine, n_nodes_total, n_samples, device, group=None, exchange="packed"):
        import torch.distributed as dist

        if exchange not in ("packed", "allreduce"):
            raise ValueError("exchange must be 'packed' or 'allreduce'")
        self.engine = engine
        self.n_nodes_total = int(n_nodes_total)
        self.n_samples = int(n_samples)
        self.group = group
        self.exchange = exchange
        self.device = torch.device(device)
        self.world = dist.get_world_size(group)
        ns = self.n_samples
        # this rank's partial: rows (max, idx bits, sum) of one buffer, so that the three series
        # travel in one message
        self.packed = torch.empty((3, ns), dtype=torch.float64, device=self.device)
        self.pmax = self.packed[0]
        self.pidx = self.packed[1].view(torch.int64)
        self.psum = self.packed[2]
        self.gathered = torch.empty((self.world, 3, ns), dtype=torch.float64, device=self.device)
        self.out = (torch.empty(ns, dtype=torch.float64, device=self.device),
                    torch.empty(ns, dtype=torch.float64, device=self.device),
                    torch.empty(ns, dtype=torch.int64, device=self.device))
        # tie_rule = 1: a rank's outcome (exp bits, index bits) and the gathered ones; every rank of the
        # group must be configured alike (the second exchange is a collective)
        self.tie_rule = _agreed_on_all_ranks(engine.get("tie_rule") if hasattr(engine, "get") else False,
                                             self.device, group)
        if self.tie_rule and exchange != "packed":
            raise ValueError("tie_rule = 1 on a sharded detect uses the packed exchange")
        self.tie_packed = torch.zeros((2, ns), dtype=torch.float64, device=self.device)
        self.tie_packed[1].view(torch.int64).fill_(INT64_MAX)
        self.tie_gathered = torch.empty((self.world, 2, ns), dtype=torch.float64, device=self.device)
        self._bound = None

    def _bind_stream(self):
        if self.device.type != "cuda":
            return
        ptr = torch.cuda.current_stream(self.device).cuda_stream
        if self._bound != ptr:
            self.engine.set_stream(ptr)
            self._bound = ptr

    def detect(self, log_onsets, fsmp, lsmp, available, out=None):
        """
        One sharded detect step; returns ``(max_coa, max_norm_coa, max_coa_idx)`` on every rank.

        The result lands in ``out`` (three device tensors of length ``n_samples``) if given, else
        in this detector's own buffers -- the SAME three tensors on every call, whichever
        exchange is used: they are valid until the next ``detect``; callers that keep results
        across steps pass ``out=`` or clone.
        """
        out = self.out if out is None else out
        self._bind_stream()
        if self.engine.n_rows is None:                       # empty slab: the neutral partial
            self.pmax.fill_(float("-inf"))
            self.pidx.fill_(INT64_MAX)
            self.psum.zero_()
        else:
            self.engine.detect_partial(log_onsets, fsmp, lsmp, available,
                                       (self.pmax, self.pidx, self.psum))
        if self.exchange == "allreduce":
            got = exchange_partials(self.pmax, self.pidx, self.psum, self.n_nodes_total,
                                    self.group)
            for dst, src in zip(out, got):
                dst.copy_(src)
            return out
        all_gather_packed(self.packed, self.gathered, self.group)
        out = self.engine.finalize_packed(self.gathered, self.world, self.n_samples,
                                          self.n_nodes_total, out=out)
        if self.tie_rule:
            if self.engine.n_rows is not None:               # (an empty slab keeps the neutral outcome)
                self.engine.tie_partial(log_onsets, fsmp, lsmp, available, self.gathered, self.world,
                                        self.tie_packed)
            all_gather_packed(self.tie_packed, self.tie_gathered, self.group)
            self.engine.tie_fold(self.tie_gathered, self.world, self.n_samples, out[2])
        return out

    def marginal_map(self, log_onsets, fsmp, lsmp, available, first_sample, end_sample, nx_total,
                     plane_shape=None):
        """
        Locate without the volume on a sharded grid: every rank marginalises its slab
        (``Engine.marginal_map``), the slabs are gathered.  Returns the whole map on every rank.
        ``plane_shape`` = ``(ny, nz)`` is needed only on a rank with an empty slab.
        """
        self._bind_stream()
        if self.engine.n_rows is None:                       # empty slab: nothing to add
            if plane_shape is None:
                raise ValueError("a rank without x-planes needs plane_shape=(ny, nz)")
            local = torch.zeros((0,) + tuple(plane_shape), dtype=torch.float64, device=self.device)
            return gather_planes(local, nx_total, self.group)
        nx, ny, nz = self.engine.grid
        local = torch.zeros((nx, ny, nz), dtype=torch.float64, device=self.device)
        self.engine.marginal_map(log_onsets, fsmp, lsmp, available, first_sample, end_sample,
                                 out=local, n_nodes_total=self.n_nodes_total)
        return gather_planes(local, nx_total, self.group)


MAX_BOXES = 3


class ColumnShardedDetector:
    """
    One rank's share of a detect sweep sharded by flat-index ranges at column granularity
    (:func:`shard_columns`): up to three engines, one per box of :func:`column_boxes`, each with
    its box resident (``load_lut(box, node_offset=(x0 * ny + y0) * nz)``).  Per step every engine
    writes its partial into one row block of a packed ``[3 boxes][3][n_samples]`` buffer (absent
    boxes keep the neutral partial), ONE all-gather moves it to every rank, and the device fold of
    ``Engine.finalize_packed`` over ``world * 3`` sets gives the global series -- the same exchange
    as :class:`ShardedDetector`, three sets per rank instead of one.  A rank without any column
    passes ``engines=[]`` and a ``fold_engine`` (an Engine without a table).

    Stream discipline as :class:`ShardedDetector`: every engine is bound to torch's current stream
    before the step.
    """

    def __init__(self, engines, n_nodes_total, n_samples, device, group=None, fold_engine=None):
        import torch.distributed as dist

        self.engines = list(engines)
        if len(self.engines) > MAX_BOXES:
            raise ValueError("a column range is at most three boxes")
        self.fold_engine = fold_engine if fold_engine is not None else (
            self.engines[0] if self.engines else None)
        if self.fold_engine is None:
            raise ValueError("a rank without columns needs a fold_engine")
        self.n_nodes_total = int(n_nodes_total)
        self.n_samples = ns = int(n_samples)
        self.group = group
        self.device = torch.device(device)
        self.world = dist.get_world_size(group)
        self.packed = torch.empty((MAX_BOXES, 3, ns), dtype=torch.float64, device=self.device)
        self.packed[:, 0] = float("-inf")                    # neutral partials
        self.packed[:, 1].view(torch.int64).fill_(INT64_MAX)
        self.packed[:, 2] = 0.0
        self.gathered = torch.empty((self.world, MAX_BOXES, 3, ns), dtype=torch.float64,
                                    device=self.device)
        self.out = (torch.empty(ns, dtype=torch.float64, device=self.device),
                    torch.empty(ns, dtype=torch.float64, device=self.device),
                    torch.empty(ns, dtype=torch.int64, device=self.device))
        # tie_rule = 1 (as ShardedDetector): one outcome per box
        rules = {bool(e.get("tie_rule")) for e in self.engines + [self.fold_engine]}
        if len(rules) > 1:
            raise ValueError("the engines of one rank must share their tie_rule")
        self.tie_rule = _agreed_on_all_ranks(rules.pop(), self.device, group)
        self.tie_packed = torch.zeros((MAX_BOXES, 2, ns), dtype=torch.float64, device=self.device)
        self.tie_packed[:, 1].view(torch.int64).fill_(INT64_MAX)
        self.tie_gathered = torch.empty((self.world, MAX_BOXES, 2, ns), dtype=torch.float64,
                                        device=self.device)
        self._bound = None

    def _bind_stream(self):
        if self.device.type != "cuda":
            return
        ptr = torch.cuda.current_stream(self.device).cuda_stream
        if self._bound != ptr:
            for eng in {id(e): e for e in self.engines + [self.fold_engine]}.values():
                eng.set_stream(ptr)
            self._bound = ptr

    def detect(self, log_onsets, fsmp, lsmp, available, out=None):
        """Returns ``(max_coa, max_norm_coa, max_coa_idx)``; in ``out`` if given, else in this
        detector's own buffers (valid until the next ``detect``)."""
        out = self.out if out is None else out
        self._bind_stream()
        for k, eng in enumerate(self.engines):
            eng.detect_partial(log_onsets, fsmp, lsmp, available,
                               (self.packed[k, 0], self.packed[k, 1].view(torch.int64),
                                self.packed[k, 2]))
        all_gather_packed(self.packed, self.gathered, self.group)
        out = self.fold_engine.finalize_packed(self.gathered, self.world * MAX_BOXES,
                                               self.n_samples, self.n_nodes_total, out=out)
        if self.tie_rule:
            for k, eng in enumerate(self.engines):
                eng.tie_partial(log_onsets, fsmp, lsmp, available, self.gathered,
                                self.world * MAX_BOXES, self.tie_packed[k])
            all_gather_packed(self.tie_packed, self.tie_gathered, self.group)
            self.fold_engine.tie_fold(self.tie_gathered, self.world * MAX_BOXES, self.n_samples, out[2])
        return out
