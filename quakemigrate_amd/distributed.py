# -*- coding: utf-8 -*-
"""
Multi-GPU sharding of the migrate / find_max_coa path.

Nodes are independent in ``migrate`` and ``find_max_coa`` is a reduction over
nodes, so the 3-D grid is sharded by contiguous ranges of x-planes (= contiguous
ranges of flat node indices, flat = (ix*ny+iy)*nz+iz, quakemigrate/lut/lut.py:165-166):
one process per GPU, each with its slab of the travel-time table resident, the
(small) onset array replicated.  Per timestep every rank produces its partial
``(log2-domain max, global argmax, sum of coalescence)`` per sample with
``Engine.detect_partial`` and the only exchange is three tiny all-reduces over
``n_samples`` elements (48 KB each at 6000 samples) -- RCCL over xGMI on the GPU
box, gloo in the CPU tests:

    gmax = all_reduce(pmax, MAX)
    gidx = all_reduce(where(pmax == gmax, pidx, INT64_MAX), MIN)   # lowest index wins
    gsum = all_reduce(psum, SUM)
    max_coa = 2**gmax;  max_norm_coa = max_coa * n_nodes_total / gsum

which reproduces the reference's tie-break (strict '>' in ascending node order,
migratelib.c:102) exactly, because equal maxima on two ranks resolve to the lower
global index.  The functions are plain torch ops on whatever device the tensors
live on -- plumbing, not the hot path.
"""

from __future__ import annotations

import torch

INT64_MAX = torch.iinfo(torch.int64).max


def shard_planes(nx: int, world_size: int, rank: int):
    """[x0, x1) of the x-planes owned by ``rank`` (balanced, contiguous)."""
    base, extra = divmod(nx, world_size)
    x0 = rank * base + min(rank, extra)
    return x0, x0 + base + (1 if rank < extra else 0)


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
    n_samples).  Returns ``(max_coa, max_norm_coa, max_coa_idx)`` on every rank.
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
    pieces = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(pieces, padded, group=group)
    return torch.cat([p[: x1 - x0] for p, (x0, x1) in zip(pieces, sizes)], dim=0)


class ShardedDetector:
    """
    One rank's share of a grid-sharded detect sweep.

    ``engine`` has this rank's slab resident (``load_lut(slab, node_offset=...)``);
    ``detect(log_onsets_dev, fsmp, lsmp, available)`` returns the global series.
    """

    def __init__(self, engine, n_nodes_total, n_samples, device, group=None):
        self.engine = engine
        self.n_nodes_total = int(n_nodes_total)
        self.group = group
        self.pmax = torch.empty(n_samples, dtype=torch.float64, device=device)
        self.psum = torch.empty(n_samples, dtype=torch.float64, device=device)
        self.pidx = torch.empty(n_samples, dtype=torch.int64, device=device)

    def detect(self, log_onsets, fsmp, lsmp, available):
        self.engine.detect_partial(log_onsets, fsmp, lsmp, available,
                                   (self.pmax, self.pidx, self.psum))
        return exchange_partials(self.pmax, self.pidx, self.psum,
                                 self.n_nodes_total, self.group)

    def marginal_map(self, log_onsets, fsmp, lsmp, available, first_sample, end_sample, nx_total):
        """
        Locate without the volume on a sharded grid: every rank marginalises its slab
        (``Engine.marginal_map``), the slabs are gathered.  Returns the whole map on every rank.
        """
        nx, ny, nz = self.engine.grid
        local = torch.zeros((nx, ny, nz), dtype=torch.float64, device=self.pmax.device)
        self.engine.marginal_map(log_onsets, fsmp, lsmp, available, first_sample, end_sample,
                                 out=local, n_nodes_total=self.n_nodes_total)
        return gather_planes(local, nx_total, self.group)
