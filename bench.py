# -*- coding: utf-8 -*-
"""
bench.py -- detect-sweep throughput of the MI355X migration engine.

Metric (BASELINE.json): grid-nodes x time-samples stacked per second on the detect
sweep.  A "step" is one fused migrate + find_max_coa pass (QuakeScan._compute's hot
path, quakemigrate/signal/scan.py:635-638) over one timestep of synthetic onsets
(BASELINE.md recipe, quakemigrate_amd/synth.py) with the travel-time table and the
log-onsets already resident in HBM.

Workload at N=1: C3 = 201x201x101 nodes x 30 onset rows x 6000 samples -- the
configuration the north-star's target is quoted on.  N>1 (launched by
torch.distributed.run, one rank per GPU): weak scaling -- every rank holds a
C3-sized slab of x-planes of a grid N times longer in x, stacks it, and the ranks
exchange their per-sample (max, argmax, sum) partials with three 48 KB RCCL
all-reduces per step (the path's only exchange).

Detect runs screened by default: a float32 sweep over every
node-sample plus an exact float64 re-evaluation of the cells that can hold the maximum
-- max_coa and max_coa_idx are the float64 kernel's bits, max_norm_coa is within ~3e-9
(contract 1e-6).  The line therefore also carries `exact_f64`: the same steps on the
float64 kernel (Engine(screen=0)), timed the same way, and the two results compared.

One JSON line on stdout (rank 0).  Besides the contract's keys it carries
  roofline      : the dominant kernel (the sweep / fused LDS-tiled stack) against HBM
                  with its ALGORITHMIC bytes (table + onsets + outputs) -- by
                  construction far below 1 %: the kernel is LDS-gather / VALU bound,
                  not HBM bound;
  roofline_onchip : the ceilings that do bind it (LDS operand bytes, VALU ops);
  screening, exact_f64 : see above;
  roofline_materialised : the locate-style variant that writes the 4-D volume
                  (8 B per node-sample of real HBM traffic), the figure the
                  north-star's ">= 50 % of HBM" maps to;
  cpu_baseline  : the reference's two loops (oracle/_ref when present, else the
                  oracle port) timed on this host's cores on a bounded sample.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12           # B/s  (MI355X_MICROARCH.md: HBM3E 8 TB/s spec)
LDS_PEAK = 150.0e12         # B/s  aggregate ds_read_b64/b128
FP64_PEAK = 39.3e12         # vector FP64 instructions-lanes / s (78.6 TFLOP/s FMA)
PMC_TRAFFIC_C3_SWEEP = (1.694e5 + 2.291e5) * 1024     # bytes per screen_lds_kernel launch at C3
PMC_TRAFFIC_C3_F64 = (1.60e5 + 3.5e4) * 1024          # bytes per stack_lds_kernel launch at C3
EXP_F32_OPS = 8             # float32 ops per node-sample besides the S adds in the screening sweep
EXP_FP64_OPS = 18           # FP64-rate VALU ops per node-sample besides the S adds (2^z, sum, max)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3",
                    help="C3 (default, weak-scaled slab per GPU), C2, C1; C4 = the 401x401x201 x "
                         "60 x 12000 grid partitioned over the N GPUs; C5 = C3 as a continuous "
                         "stream of --steps timesteps with copies overlapped on HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-materialised", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--engine", default="{}", help="json dict of engine tunables")
    ap.add_argument("--no-exact", action="store_true",
                    help="skip the float64-kernel comparison run (exact_f64)")
    ap.add_argument("--materialise-full", action="store_true",
                    help="also run configs[2] literally: the whole n_samples volume (196 GB at "
                         "C3) written to HBM with the scan outputs (needs the memory)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="single process: build and time the slab rank --emulate-rank of an "
                         "N-GPU partition would hold (no exchange); for measuring C4 on 1 GPU")
    ap.add_argument("--emulate-rank", type=int, default=0)
    return ap.parse_args()


def cpu_baseline(case, budget_s):
    """Reference loops on the host cores over a bounded time-chunk of the workload."""
    from oracle import qm_oracle as oq

    impl = "ref" if oq.have_ref() else "port"
    fns = oq._ref() if impl == "ref" else oq._port()
    threads = os.cpu_count() or 1
    lon = np.ascontiguousarray(oq.log_onsets(case.onsets))
    tt = np.ascontiguousarray(case.traveltimes)
    n_nodes = int(np.prod(tt.shape[:-1]))
    rows, t_samples = lon.shape

    def run(ns, prefault):
        vol = np.zeros((n_nodes, ns))
        if prefault:
            vol.fill(1.0)
            vol.fill(0.0)
        a, b, c = np.zeros(ns), np.zeros(ns), np.zeros(ns, dtype=np.int64)
        t0 = time.perf_counter()
        fns["stack"](lon, tt, vol, case.fsmp, t_samples - case.fsmp - ns, ns, rows,
                     case.available, n_nodes, threads)
        t1 = time.perf_counter()
        fns["scan"](vol, a, b, c, ns, n_nodes, threads)
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1

    ns = 8
    m, s = run(ns, True)                               # rate probe (and library warm-up)
    rate = n_nodes * ns / max(m + s, 1e-6)
    max_ns_mem = max(8, int((4 << 30) // (8 * n_nodes)))
    ns = int(min(case.n_samples, max_ns_mem, max(8, rate * budget_s / 2.5 / n_nodes)))
    warm = run(ns, True)
    reps = 1
    while sum(warm) < budget_s / 2 and reps < 16:      # accumulate a bounded amount of work
        w = run(ns, True)
        warm = (warm[0] + w[0], warm[1] + w[1])
        reps += 1
    warm = (warm[0] / reps, warm[1] / reps)
    cold = run(ns, False)                              # fresh np.zeros: the reference's case
    work = n_nodes * ns
    return {
        "value": work / sum(warm), "unit": "node-samples/s", "cores": threads,
        "kind": "reference" if impl == "ref" else "port",
        "sample": f"{case.name} grid {tuple(tt.shape[:-1])} x {rows} rows, {ns} of "
                  f"{case.n_samples} samples (time chunk) x {reps} repeats, "
                  f"migrate+find_max_coa, volume pre-faulted",
        "migrate_s": warm[0], "find_max_coa_s": warm[1],
        "cold_value": work / sum(cold),
    }


def onchip(screened, local_ns, S, kern_s):
    """The ceilings that bind the stacking kernel: LDS operand bytes and VALU issue."""
    if screened:
        # 4 operand bytes per add; packed float32: two adds per lane-instruction
        ops = S + EXP_F32_OPS
        return {"lds": {"achieved": 4.0 * local_ns * S / kern_s / 1e12, "peak": LDS_PEAK / 1e12,
                        "unit": "TB/s", "frac": 4.0 * local_ns * S / kern_s / LDS_PEAK},
                "fp32_packed_valu": {"achieved": local_ns * ops / kern_s / 1e12,
                                     "peak": 2 * FP64_PEAK / 1e12, "unit": "Tops/s",
                                     "frac": local_ns * ops / kern_s / (2 * FP64_PEAK),
                                     "ops_per_node_sample": ops}}
    return {"lds": {"achieved": 8.0 * local_ns * S / kern_s / 1e12, "peak": LDS_PEAK / 1e12,
                    "unit": "TB/s", "frac": 8.0 * local_ns * S / kern_s / LDS_PEAK},
            "fp64_valu": {"achieved": local_ns * (S + EXP_FP64_OPS) / kern_s / 1e12,
                          "peak": FP64_PEAK / 1e12, "unit": "Tinstr-lanes/s",
                          "frac": local_ns * (S + EXP_FP64_OPS) / kern_s / FP64_PEAK,
                          "ops_per_node_sample": S + EXP_FP64_OPS}}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    from quakemigrate_amd import distributed as qd
    from quakemigrate_amd import synth
    from quakemigrate_amd.core import lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # development aid: QM_BENCH_ONE_DEVICE=1 runs the N>1 code path with every rank on GPU 0 and
    # the gloo backend (RCCL refuses two ranks on one device) -- plumbing check on a 1-GPU box
    one_device = os.environ.get("QM_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif args.gpus > 1:
        raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py ...")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    part_world, part_rank = world, rank                 # partition used to cut the grid
    if args.emulate_world > 0 and world == 1:
        part_world, part_rank = args.emulate_world, args.emulate_rank

    # ---- workload: this rank's slab of the (weak-scaled) grid -----------------------
    streaming = args.config == "C5"
    cfg_name = "C3" if streaming else args.config
    base = synth.CONFIGS[cfg_name]
    nx, ny, nz = base["grid"]
    if cfg_name == "C4":                                # fixed grid, partitioned over the GPUs
        grid = (nx, ny, nz)
        x_range = qd.shard_planes(nx, part_world, part_rank)
    else:                                               # weak scaling: a full slab per GPU
        grid = (nx * part_world, ny, nz)
        x_range = (nx * part_rank, nx * (part_rank + 1))
    n_pool = 3                                          # distinct timesteps cycled through
    cases = [synth.make_case(cfg_name, step=s, grid=grid, x_range=x_range)
             for s in range(n_pool)]
    case = cases[0]
    S, ns = case.available, case.n_samples
    n_total = case.n_nodes_total
    n_local = int(np.prod(case.traveltimes.shape[:-1]))
    t_samples = case.onsets.shape[1]

    tunables = {}                  # brick shape, samples per lane, workgroup layout: per table
    tunables.update(json.loads(args.engine))
    eng = lib.Engine(local_rank, **tunables)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.load_lut(case.traveltimes, node_offset=x_range[0] * ny * nz)
    assert eng.lut_max <= case.lsmp
    onsets_dev = [torch.from_numpy(np.ascontiguousarray(
        np.log(np.clip(c.onsets, 0.01, np.inf)))).to(dev) for c in cases]
    out = (torch.empty(ns, dtype=torch.float64, device=dev),
           torch.empty(ns, dtype=torch.float64, device=dev),
           torch.empty(ns, dtype=torch.int64, device=dev))
    sharded = qd.ShardedDetector(eng, n_total, ns, dev) if world > 1 else None

    def step(i):
        on = onsets_dev[i % n_pool]
        if sharded is None:
            eng.detect(on, case.fsmp, case.lsmp, case.available, n_nodes_total=n_total,
                       out=out)
            return out
        return sharded.detect(on, case.fsmp, case.lsmp, case.available)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    res = step(0)          # set-up, not a warmup step: sizes the scratch buffers, builds the
    fence()                # sweep's offset table for this scan length (once per table)
    for i in range(args.warmup):
        res = step(i)
    if streaming:
        # host onset windows -> pinned -> H2D on a copy stream, detect + D2H on a compute
        # stream; the timed region includes every copy (PCIe-inclusive rate).  The pipeline's
        # pinned / device buffers are set up once, before the clock starts.
        from quakemigrate_amd.stream import StreamingDetector

        host = [np.ascontiguousarray(np.log(np.clip(c.onsets, 0.01, np.inf))) for c in cases]
        sd = StreamingDetector(eng, S, t_samples, case.fsmp, case.lsmp, case.available,
                               n_nodes_total=n_total, depth=3, device=dev)
    fence()
    eng.config("log_timing", 1)
    t0 = time.perf_counter()
    if streaming:
        got = sd.run(host[(args.warmup + i) % n_pool] for i in range(args.steps))
        res = tuple(torch.from_numpy(a) for a in got[-1])
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
    else:
        for i in range(args.steps):
            res = step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_calls = eng.kernel_log()
    eng.config("log_timing", 0)
    screened = eng.get("screened_steps") > 0 and eng.get("fallback_steps") == 0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- in-bench sanity: injected events are found where they were put ------------
    last = (args.warmup + args.steps - 1) % n_pool
    idx = res[2].cpu().numpy()
    for (ijk, t_ev) in cases[last].event_nodes:
        if part_world != world and not (x_range[0] <= ijk[0] < x_range[1]):
            continue                                    # emulated slab: event lies elsewhere
        found = np.unravel_index(int(idx[t_ev]), grid)
        # on coarse grids (C2-C4) this is the event's node itself; on the 25 m Icequake-sized
        # grid neighbouring nodes are within a sample of each other and the reference, too,
        # lands one node off (tests/golden/c1_icequake_geometry.npz)
        assert max(abs(int(a) - int(b)) for a, b in zip(found, ijk)) <= 2, \
            f"event at sample {t_ev}: node {found} is not at {ijk}"

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    n_norm = n_total                                    # what the engine normalises by
    if part_world != world:
        n_total = n_local                               # emulated slab: count what was stacked
    work_step = n_total * ns                            # node-samples per step, whole job
    value = work_step * args.steps / elapsed
    kern_s = kern_ms / 1e3 / max(kern_calls, 1)         # avg stacking-kernel time, this rank
    local_ns = n_local * ns
    b_fused = 4.0 * n_local * S + 8.0 * S * t_samples + 24.0 * ns   # SURVEY 8d B_F
    result = {
        "metric": "grid-nodes x time-samples stacked /sec (detect sweep)",
        "value": value, "unit": "node-samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 sweep + f64 refinement" if screened else "f64", "data": "synthetic",
        "config": {"workload": f"{args.config} detect sweep: {x_range[1] - x_range[0]}x{ny}x{nz} "
                               f"nodes on rank 0 (grid {grid[0]}x{ny}x{nz}), {S} onset rows, "
                               f"{ns} samples per step, fused migrate+find_max_coa, table "
                               f"resident in HBM, onsets "
                               + ("streamed from pinned host memory (copies inside the timed "
                                  "region)" if streaming else "resident in HBM"),
                   "n_nodes_per_gpu": n_local, "n_rows": S, "n_samples": ns,
                   "sharding": "x-planes" if world > 1 else "none",
                   "exchange": "3 x all_reduce(n_samples) per step (RCCL)" if world > 1
                   else "none", "engine": dict(tunables, brick=[eng.get("brick_x"),
                                                                   eng.get("brick_y"),
                                                                   eng.get("brick_z")],
                                               samples_per_lane=eng.get("samples_per_lane"),
                                               waves=eng.get("waves"))},
        "kernel": {"name": (f"qm::screen_lds_kernel<{eng.get('screen_pairs')},{(S + 7) // 8}>" if screened
                            else f"qm::stack_lds_kernel<{eng.get('samples_per_lane')},false,"
                                 f"{(S + 7) // 8}>"), "avg_ms": kern_s * 1e3,
                   "launches": kern_calls, "timing": "HIP events on the launch stream"},
        "roofline": {"bound": "hbm", "achieved": b_fused / kern_s / 1e9,
                     "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": b_fused / kern_s / HBM_PEAK,
                     # PMC FETCH_SIZE + WRITE_SIZE of one launch (separate --pmc passes,
                     # profiles/r01_pmc_C3_{fetch,write}.csv); other configs: not collected
                     "traffic": (PMC_TRAFFIC_C3_SWEEP if screened else PMC_TRAFFIC_C3_F64)
                     if (cfg_name == "C3" and world == 1) else None,
                     "algorithmic_bytes_per_launch": b_fused,
                     "note": "fused detect never writes the volume: compulsory HBM bytes "
                             "are the table, the onsets and the outputs only; the kernel "
                             "is LDS-gather / VALU bound (see roofline_onchip)"},
        "roofline_onchip": onchip(screened, local_ns, S, kern_s),
    }
    if screened:
        result["screening"] = {
            "what": "every node-sample is stacked in float32 (pairs of samples: ds_read_b64 + "
                    "v_pk_add_f32); every (brick, sample) cell within the rigorous float32 "
                    "error bound of the sample's maximum is re-evaluated node by node in "
                    "float64: max_coa and max_coa_idx are those of the float64 kernel, "
                    "max_norm_coa's sum over nodes has float32 terms",
            "candidate_cells_last_step": eng.get("last_candidates"),
            "steps_screened": eng.get("screened_steps"),
            "steps_fallen_back_to_float64": eng.get("fallback_steps")}

    # ---- the float64 kernel on the same steps, and the two paths side by side --------
    if screened and world == 1 and not streaming and not args.no_exact:
        ex = lib.Engine(local_rank, **dict(tunables, screen=0))
        ex.set_stream(torch.cuda.current_stream().cuda_stream)
        ex.load_lut(case.traveltimes, node_offset=x_range[0] * ny * nz)
        out_x = tuple(torch.empty_like(o) for o in out)
        n_x = max(2, min(args.steps, 10))
        ex.detect(onsets_dev[last], case.fsmp, case.lsmp, case.available, n_nodes_total=n_norm,
                  out=out_x)
        torch.cuda.synchronize()
        same_idx = bool(torch.equal(out_x[2], res[2]))
        same_coa = bool(torch.equal(out_x[0], res[0]))
        norm_rel = float(((out_x[1] - res[1]).abs() / out_x[1]).max().item())
        assert same_idx and same_coa and norm_rel < 1e-6, (same_idx, same_coa, norm_rel)
        ex.config("log_timing", 1)
        t0 = time.perf_counter()
        for i in range(n_x):
            ex.detect(onsets_dev[i % n_pool], case.fsmp, case.lsmp, case.available,
                      n_nodes_total=n_norm, out=out_x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_x
        xk_ms, xk_calls = ex.kernel_log()
        xk_s = xk_ms / 1e3 / max(xk_calls, 1)
        result["exact_f64"] = {
            "ms_per_step": dt * 1e3, "value": work_step / dt, "unit": "node-samples/s",
            "steps": n_x,
            "kernel": {"name": f"qm::stack_lds_kernel<{ex.get('samples_per_lane')},false,"
                               f"{(S + 7) // 8}>", "avg_ms": xk_s * 1e3, "launches": xk_calls},
            "roofline_onchip": onchip(False, local_ns, S, xk_s),
            "screened_vs_float64": {"max_coa_idx_identical": same_idx,
                                    "max_coa_identical": same_coa,
                                    "max_norm_coa_max_rel_diff": norm_rel}}
        ex.close()

    # ---- locate-style materialising variant on the same grid (HBM-write bound) ------
    if not args.no_materialised and world == 1 and cfg_name == "C3" and not streaming:
        ns_loc = 401                                    # 4 * marginal_window(2 s) * 50 Hz + 1
        on = onsets_dev[0][:, : case.fsmp + ns_loc + case.lsmp].contiguous()
        vol = torch.empty((n_local, ns_loc), dtype=torch.float64, device=dev)
        o2 = tuple(torch.empty(ns_loc, dtype=d, device=dev)
                   for d in (torch.float64, torch.float64, torch.int64))
        eng.migrate(on, case.fsmp, case.lsmp, case.available, vol, scan_out=o2)
        torch.cuda.synchronize()
        eng.config("log_timing", 1)
        reps = 5
        for _ in range(reps):
            eng.migrate(on, case.fsmp, case.lsmp, case.available, vol, scan_out=o2)
        torch.cuda.synchronize()
        ms, calls = eng.kernel_log()
        eng.config("log_timing", 0)
        sec = ms / 1e3 / calls
        b_mat = 8.0 * n_local * ns_loc + 4.0 * n_local * S + \
            8.0 * S * on.shape[1] + 24.0 * ns_loc      # SURVEY 8d figure 1
        result["roofline_materialised"] = {
            "bound": "hbm", "achieved": b_mat / sec / 1e9, "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": b_mat / sec / HBM_PEAK, "avg_ms": sec * 1e3,
            "node_samples_per_s": n_local * ns_loc / sec,
            "workload": f"locate window: same grid, {ns_loc} samples, volume "
                        f"({8.0 * n_local * ns_loc / 1e9:.1f} GB) written to HBM + scan"}
        # locate without the volume: marginalised 3-D map over the central half of the window
        cmap = torch.empty(n_local, dtype=torch.float64, device=dev)
        eng.marginal_map(on, case.fsmp, case.lsmp, case.available, 100, 301, out=cmap,
                         scan_out=o2)
        torch.cuda.synchronize()
        eng.config("log_timing", 1)
        for _ in range(reps):
            eng.marginal_map(on, case.fsmp, case.lsmp, case.available, 100, 301, out=cmap,
                             scan_out=o2)
        torch.cuda.synchronize()
        ms, calls = eng.kernel_log()
        eng.config("log_timing", 0)
        result["locate_marginal"] = {
            "avg_ms": ms / calls, "node_samples_per_s": n_local * ns_loc / (ms / 1e3 / calls),
            "workload": "same window, marginalised map (sum over samples 100..300) + scan; "
                        "no n_nodes x n_samples store"}
        # find_max_coa alone on that resident volume (scan_volume_kernel): pure HBM read
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        eng.find_max_coa(vol, ns_loc, n_local, o2)
        e0.record()
        for _ in range(reps):
            eng.find_max_coa(vol, ns_loc, n_local, o2)
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) / 1e3 / reps
        result["roofline_find_max_coa"] = {
            "bound": "hbm", "achieved": 8.0 * n_local * ns_loc / sec / 1e9, "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": 8.0 * n_local * ns_loc / sec / HBM_PEAK, "avg_ms": sec * 1e3,
            "workload": "find_max_coa of the resident locate volume (scan + combine kernels)"}
        del vol

    if args.materialise_full and world == 1 and not streaming:
        vol = torch.empty((n_local, ns), dtype=torch.float64, device=dev)
        o3 = tuple(torch.empty(ns, dtype=d, device=dev)
                   for d in (torch.float64, torch.float64, torch.int64))
        eng.migrate(onsets_dev[0], case.fsmp, case.lsmp, case.available, vol, scan_out=o3)
        torch.cuda.synchronize()
        eng.config("log_timing", 1)
        for _ in range(3):
            eng.migrate(onsets_dev[0], case.fsmp, case.lsmp, case.available, vol, scan_out=o3)
        torch.cuda.synchronize()
        ms, calls = eng.kernel_log()
        eng.config("log_timing", 0)
        sec = ms / 1e3 / calls
        b_full = 8.0 * n_local * ns + b_fused
        eng.detect(onsets_dev[0], case.fsmp, case.lsmp, case.available, n_nodes_total=n_total,
                   out=out)
        torch.cuda.synchronize()
        result["roofline_materialised_full"] = {
            "bound": "hbm", "achieved": b_full / sec / 1e9, "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": b_full / sec / HBM_PEAK, "avg_ms": sec * 1e3,
            "node_samples_per_s": n_local * ns / sec,
            "scan_equals_fused_detect": bool(torch.equal(o3[2], out[2])
                                             and torch.equal(o3[0], out[0])),
            "workload": f"whole step materialised: {8.0 * n_local * ns / 1e9:.0f} GB volume + scan"}
        del vol

    if not args.no_cpu_baseline and world == 1 and not streaming:
        result["cpu_baseline"] = cpu_baseline(case, args.cpu_seconds)
    else:
        result["cpu_baseline"] = None
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
