# -*- coding: utf-8 -*-
"""
bench.py -- detect-sweep throughput of the MI355X migration engine.

Metric (BASELINE.json): grid-nodes x time-samples stacked per second on the detect
sweep.  A "step" is one fused migrate + find_max_coa pass (QuakeScan._compute's hot
path, quakemigrate/signal/scan.py:635-638) over one timestep of synthetic onsets
(BASELINE.md recipe, quakemigrate_amd/synth.py) with the travel-time table and the
log-onsets already resident in HBM when the clock starts.

Precision: the headline is the reference's own -- float64 end to end
(``Engine`` default, ``dtype: "f64"``): float64 sums bit-identical to migratelib.c's,
float64 2^z, exact argmax.  The opt-in screened detect (``Engine(screen=1)``: float32
sweep + exact float64 refinement of the cells that can hold the maximum) is timed on the
same steps and reported under the extra key ``screened_optin`` -- never as ``value``.

Workload at N=1: C3 = 201x201x101 nodes x 30 onset rows x 6000 samples -- the
configuration the north-star's target is quoted on.  N>1: the SAME C3 sweep with the
grid sharded over the N GPUs by contiguous x-plane slabs (``"scaling": "strong"``: total
work fixed, so value(N) / value(1) is the speed-up), one rank per GPU, the ranks exchanging
their per-sample (max, argmax, sum) partials with ONE packed RCCL all-gather per step (the
path's only exchange).  ``--config C4`` partitions the
401x401x201 x 60 x 12000 grid of BASELINE configs[3] the same way; ``--weak`` gives every
rank a full C3-sized slab instead.  ``python bench.py --gpus N`` launches its own
``torch.distributed.run`` when it is not already running under one.

One JSON line on stdout (rank 0).  Besides the contract's keys it carries
  roofline      : the dominant kernel (fused LDS-tiled float64 stack) against HBM with its
                  ALGORITHMIC bytes (table + onsets + outputs) -- by construction far below
                  1 %: the kernel is LDS-gather / VALU bound, not HBM bound (SURVEY.md
                  section 8d) -- plus ``lds_frac`` / ``fp64_valu_frac``, the ceilings that do
                  bind it, and ``adds_only_frac`` (the reference's S adds alone); ``traffic`` =
                  HBM bytes per launch from the newest stored PMC passes over this kernel and
                  workload (profiles/rNN_pmc_traffic.json; null if none matches);
  step_with_copies : the same steps fed from pinned host memory -- H2D of the onsets and D2H
                  of the three series inside the timed region, overlapped on HIP streams
                  (SURVEY.md section 8d's PCIe-inclusive step; never ``value``);
  screened_optin : see above;
  tie_rule_optin : the same steps on Engine(tie_rule=1) -- the reference's arg-max rule on near-ties
                  (migratelib.c:98-105), opt-in -- beside the default engine: ms per step, ratio,
                  identical series on this (generic) data;
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
EXP_F32_OPS = 8             # 32-bit ops per node-sample besides the S adds in the screening sweep
EXP_FP64_OPS = 19           # FP64-rate VALU ops per node-sample besides the S adds (2^z, sum, max,
                            # group merge / address: the shift-reuse kernel issues 19.2, the round-2
                            # exact kernels 18 + 1 address add per row)
EXP_FP64_OPS_LAZY = 17      # ... the shift-reuse kernel's lazy arg-max flavour (one max instead of
                            # compare + select + max per node-sample, z folded into two FMAs; recovery
                            # and merge where a group reaches the running maximum;
                            # profiles/r03_pmc_C3shift_*: SQ_INSTS_VALU per computed node-sample - S)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3",
                    help="C3 (default), C2, C1, C4 (401x401x201 x 60 rows x 12000 samples), E1 / E2 (the "
                         "sizes the reference's Volcanotectonic / Askja examples run detect at): the "
                         "grid is partitioned over the N GPUs by x-plane slabs; C5 = C3 as a "
                         "continuous stream of --steps timesteps with copies overlapped on HIP "
                         "streams inside the timed region")
    ap.add_argument("--weak", action="store_true",
                    help="N>1: weak scaling -- every rank holds a full slab of --config's size "
                         "(grid N times longer in x) instead of 1/N of it")
    ap.add_argument("--partition", default="columns", choices=["columns", "planes"],
                    help="N>1: contiguous flat-index ranges cut at (x, y)-column granularity "
                         "(default: balanced to one column of nz nodes; a rank holds up to three "
                         "boxes) or whole x-planes (201 planes over 8 ranks = 26 / 25)")
    ap.add_argument("--exchange", default="packed", choices=["packed", "allreduce"],
                    help="N>1: one packed all-gather + device fold (default) or three all-reduces")
    ap.add_argument("--no-screened", action="store_true",
                    help="skip the opt-in screened detect comparison run (screened_optin)")
    ap.add_argument("--no-copies", action="store_true",
                    help="skip the PCIe-inclusive leg (step_with_copies)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-materialised", action="store_true")
    ap.add_argument("--no-table-switch", action="store_true",
                    help="skip the availability-change leg (table_switch)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--engine", default="{}", help="json dict of engine tunables")
    ap.add_argument("--no-materialise-full", action="store_true",
                    help="skip the literal BASELINE configs[2] leg (roofline_materialised_full)")
    ap.add_argument("--materialise-full", action="store_true",
                    help="also run configs[2] literally: the whole n_samples volume (196 GB at "
                         "C3) written to HBM with the scan outputs (needs the memory)")
    ap.add_argument("--steps-per-launch", type=int, default=1,
                    help="timesteps stacked by ONE launch (Engine.detect_batch; N = 1 and the C5 "
                         "stream): what fills the GPU on the reference's example-sized grids "
                         "(C1, E1, E2); --steps is rounded up to a multiple of it")
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


def onchip(screened, local_ns, S, kern_s, operands_per_add=1.0, exp_ops=EXP_FP64_OPS):
    """The ceilings that bind the stacking kernel: LDS operand bytes and VALU issue.
    ``operands_per_add``: 8-byte LDS operands fetched per add -- 1 for the kernels that read every
    operand from LDS, ~0.55 for the shift-reuse kernel (register windows shared by 8 nodes)."""
    if screened:
        # 4 operand bytes per add; v_add3_u32: two adds per lane-instruction
        ops = S + EXP_F32_OPS
        return {"lds": {"achieved": 4.0 * local_ns * S / kern_s / 1e12, "peak": LDS_PEAK / 1e12,
                        "unit": "TB/s", "frac": 4.0 * local_ns * S / kern_s / LDS_PEAK},
                "int32_valu": {"achieved": local_ns * ops / kern_s / 1e12,
                                     "peak": 2 * FP64_PEAK / 1e12, "unit": "Tops/s",
                                     "frac": local_ns * ops / kern_s / (2 * FP64_PEAK),
                                     "ops_per_node_sample": ops}}
    lds_bytes = 8.0 * local_ns * S * operands_per_add
    return {"lds": {"achieved": lds_bytes / kern_s / 1e12, "peak": LDS_PEAK / 1e12,
                    "unit": "TB/s", "frac": lds_bytes / kern_s / LDS_PEAK,
                    "operands_per_add": operands_per_add},
            "fp64_valu": {"achieved": local_ns * (S + exp_ops) / kern_s / 1e12,
                          "peak": FP64_PEAK / 1e12,
                          "unit": "TFLOP/s (FP64 VALU instruction-lanes, one operation per instruction)",
                          "frac": local_ns * (S + exp_ops) / kern_s / FP64_PEAK,
                          "ops_per_node_sample": S + exp_ops}}


def stored_traffic(label, kernel_name, algorithmic_bytes):
    """
    HBM bytes per launch of the dominant kernel from the newest stored PMC passes
    (profiles/rNN_pmc_traffic.json, written by tools/pmc_traffic.py from separate FETCH_SIZE /
    WRITE_SIZE rocprofv3 runs of this workload): PMC cannot be collected inside an un-profiled
    bench run.  Returned only when the stored kernel is the one this run launched.
    """
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        stored = json.load(open(files[-1]))
        rec = stored.get(label)
    except (OSError, ValueError):
        return None, None
    if not rec or rec.get("kernel") != kernel_name:
        return None, None
    # the counts belong to the code they were measured on: the stacking kernels' sources must be the
    # ones this run was built from (tools/pmc_traffic.py: kernel_code_digest), else nothing is reported
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_traffic
        if stored.get("_kernel_code") != pmc_traffic.kernel_code_digest():
            return None, {"stale": "profiles/" + os.path.basename(files[-1]) + " was measured on other kernel "
                                   "sources than this tree's (tools/pmc_traffic.py); not reported"}
    finally:
        sys.path.pop(0)
    total = rec["fetch_bytes"] + rec["write_bytes"]
    detail = {"fetch_bytes": rec["fetch_bytes"], "write_bytes": rec["write_bytes"],
              "fetch_bytes_if_128B_requests_tallied_at_64B": rec["fetch_bytes_upper"],
              "algorithmic_bytes": algorithmic_bytes,
              "ratio_to_algorithmic": total / algorithmic_bytes,
              "ratio_to_algorithmic_upper": (rec["fetch_bytes_upper"] + rec["write_bytes"]) / algorithmic_bytes,
              "source": "profiles/" + os.path.basename(files[-1]) + " <- " + ", ".join(rec["source"]),
              "measured_in_this_run": False, "kernel_code": stored.get("_kernel_code"),
              "note": "separate rocprofv3 --pmc passes over one launch of this kernel on this workload "
                      "(FETCH_SIZE, WRITE_SIZE: KB); on gfx950 FETCH_SIZE counts the 128-byte requests "
                      "of wide reads at 64 bytes (MI355X_MICROARCH.md), hence the upper figure; beyond "
                      "the table the kernel reads its record stream (4 S bytes per node) once per step"}
    return total, detail


def stack_kernel_name(eng, S, volume=False):
    """Name of the stacking kernel the engine's last launch used (as rocprofv3 prints it)."""
    kind, j = eng.get("last_kernel"), eng.get("last_kernel_j")
    v = "true" if volume else "false"
    if kind == 3 and eng.get("shift_row_blocks") > 1:          # tables of more than 64 rows
        return f"void qm::stack_shift_rows2_kernel<{v}, 8>"
    if kind == 3:                                              # <mode: 0 detect, 1 volume, 2 marginal map; waves>
        return f"void qm::stack_shift_kernel<{1 if volume else 0}, {eng.get('shift_waves')}>"
    if kind == 2:
        return f"qm::stack_pair_kernel<{j // 2}, {v}, {S}>"
    if kind == 1:
        return f"qm::stack_exact_kernel<{j}, {v}, {S}>"
    return f"qm::stack_lds_kernel<{j}, {v}, {(S + 7) // 8}>"


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # stdout carries the result line and nothing else: RCCL writes a version banner to the C-level
    # stdout of every process that creates a communicator (seen with RCCL 2.26.6: five lines,
    # flushed at exit, i.e. AFTER the JSON line).  File descriptor 1 is pointed at stderr for the
    # whole run -- on every rank -- and rank 0 writes its one line to the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
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
    # development aid: QM_BENCH_FORCE_DIST=1 at N = 1 runs the step through the N>1 code path on a
    # one-rank RCCL group (ShardedDetector, device all-gather, barrier / all-reduce of the timing)
    force_dist = world == 1 and os.environ.get("QM_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    backend = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if force_dist:
            import socket

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            dist.init_process_group("nccl", rank=0, world_size=1,
                                    device_id=torch.device("cuda", local_rank))
        elif one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        backend = str(dist.get_backend()).lower()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    part_world, part_rank = world, rank                 # partition used to cut the grid
    if args.emulate_world > 0 and world == 1:
        part_world, part_rank = args.emulate_world, args.emulate_rank

    # ---- workload: this rank's slab of the grid ---------------------------------------
    streaming = args.config == "C5"
    cfg_name = "C3" if streaming else args.config
    # The continuous stream (configs[4]) shards over TIME, not over the grid: timesteps are
    # independent given their onsets, so rank r scans timesteps r, r + N, ... of the WHOLE grid
    # (every rank holds the whole table: 0.5 GB at C3) and there is no collective in the data
    # path at all.  Per-GPU work is fixed as N grows: weak scaling.
    time_sharded = streaming and world > 1
    if time_sharded:
        part_world, part_rank = 1, 0
    base = synth.CONFIGS[cfg_name]
    nx, ny, nz = base["grid"]
    if args.weak:                                       # a full slab per GPU
        grid = (nx * part_world, ny, nz)
        x_range = (nx * part_rank, nx * (part_rank + 1))
    else:                                               # fixed grid, partitioned over the GPUs
        grid = (nx, ny, nz)
        x_range = qd.shard_planes(nx, part_world, part_rank)
    # column partition (the default at N > 1): this rank's flat range as up to three boxes, the
    # whole planes first in `boxes` (that engine is the one the kernel line describes)
    by_columns = (world > 1 and not args.weak and not time_sharded and part_world == world
                  and args.partition == "columns" and args.exchange == "packed")
    boxes = []
    if by_columns:
        boxes = qd.column_boxes(*qd.shard_columns(nx, ny, world, rank), ny)
        boxes.sort(key=lambda b: -(b[1] - b[0]) * (b[3] - b[2]))
        x_range = (boxes[0][0], boxes[0][1]) if boxes else (0, 0)
    n_pool = 3                                          # distinct timesteps cycled through
    cases = [synth.make_case(cfg_name, step=s, grid=grid, x_range=x_range,
                             table=(s == 0 and x_range[1] > x_range[0]))
             for s in range(n_pool)]
    case = cases[0]
    S, ns = case.available, case.n_samples
    n_total = case.n_nodes_total
    t_samples = case.onsets.shape[1]

    tunables = {}                  # brick shape, samples per lane, workgroup layout: per table
    tunables.update(json.loads(args.engine))
    eng = lib.Engine(local_rank, **tunables)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    engines = []
    if by_columns:
        for k, (bx0, bx1, by0, by1) in enumerate(boxes):
            tt = case.traveltimes if k == 0 else synth.make_case(
                cfg_name, step=0, grid=grid, x_range=(bx0, bx1)).traveltimes
            tt = np.ascontiguousarray(tt[:, by0:by1])
            e_k = eng if k == 0 else lib.Engine(local_rank, **tunables)
            e_k.set_stream(torch.cuda.current_stream().cuda_stream)
            e_k.load_lut(tt, node_offset=(bx0 * ny + by0) * nz)
            assert e_k.lut_max <= case.lsmp
            engines.append(e_k)
        n_local = sum((b[1] - b[0]) * (b[3] - b[2]) for b in boxes) * nz
    else:
        n_local = int(np.prod(case.traveltimes.shape[:-1]))
        eng.load_lut(case.traveltimes, node_offset=x_range[0] * ny * nz)
        assert eng.lut_max <= case.lsmp
    host_onsets = [np.ascontiguousarray(np.log(np.clip(c.onsets, 0.01, np.inf))) for c in cases]
    onsets_dev = [torch.from_numpy(h).to(dev) for h in host_onsets]
    out = (torch.empty(ns, dtype=torch.float64, device=dev),
           torch.empty(ns, dtype=torch.float64, device=dev),
           torch.empty(ns, dtype=torch.int64, device=dev))
    if by_columns:
        sharded = qd.ColumnShardedDetector(engines, n_total, ns, dev, fold_engine=eng)
    else:
        sharded = (qd.ShardedDetector(eng, n_total, ns, dev, exchange=args.exchange)
                   if use_dist and not time_sharded else None)

    spl = max(1, args.steps_per_launch) if (sharded is None and world == 1) else 1
    if spl > 1:
        args.steps = -(-args.steps // spl) * spl
        if not streaming:
            # the launch's K onset arrays, resident like the single step's: [K][rows][T]
            batch_dev = [torch.stack([onsets_dev[(b + j) % n_pool] for j in range(spl)]).contiguous()
                         for b in range(n_pool)]
            out_batch = (torch.empty((spl, ns), dtype=torch.float64, device=dev),
                         torch.empty((spl, ns), dtype=torch.float64, device=dev),
                         torch.empty((spl, ns), dtype=torch.int64, device=dev))

    def step(i):
        on = onsets_dev[i % n_pool]
        if spl > 1 and not streaming:                   # steps i .. i + spl - 1 in one launch
            eng.detect_batch(batch_dev[i % n_pool], case.fsmp, case.lsmp, case.available,
                             n_nodes_total=n_total, out=out_batch)
            return tuple(t[-1] for t in out_batch)
        if sharded is None:
            eng.detect(on, case.fsmp, case.lsmp, case.available, n_nodes_total=n_total,
                       out=out)
            return out
        return sharded.detect(on, case.fsmp, case.lsmp, case.available)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    res = step(0)          # set-up, not a warmup step: sizes the scratch buffers (once per table)
    fence()
    for i in range(args.warmup):
        res = step(i)
    if streaming:
        # host onset windows -> pinned -> H2D on a copy stream, detect + D2H on a compute
        # stream; the timed region includes every copy (PCIe-inclusive rate).  The pipeline's
        # pinned / device buffers are set up once, before the clock starts.
        from quakemigrate_amd.stream import StreamingDetector

        sd = StreamingDetector(eng, S, t_samples, case.fsmp, case.lsmp, case.available,
                               n_nodes_total=n_total, depth=3, device=dev, steps_per_launch=spl)
    fence()
    eng.config("log_timing", 1)
    t0 = time.perf_counter()
    if streaming:
        # (time-sharded: this rank's timesteps are rank, rank + world, ... of the stream)
        got = sd.run(host_onsets[(args.warmup + rank + world * i) % n_pool]
                     for i in range(args.steps))
        res = tuple(torch.from_numpy(a) for a in got[-1])
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
    else:
        for i in range(0, args.steps, spl):
            res = step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_calls = eng.kernel_log()
    eng.config("log_timing", 0)
    for e_k in engines[1:]:                             # the partly owned planes' engines
        e_k.config("log_timing", 0)
    screened = eng.get("screened_steps") > 0 and eng.get("fallback_steps") == 0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- in-bench sanity: injected events are found where they were put ------------
    last = (args.warmup + args.steps - 1) % n_pool
    if spl > 1 and not streaming:                       # the last launch's last step
        last = (args.warmup + args.steps - spl + spl - 1) % n_pool
    if time_sharded:
        last = (args.warmup + rank + world * (args.steps - 1)) % n_pool
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

    # where the imbalance is: every rank's average stacking-kernel time
    world_kernel_ms = [kern_ms / max(kern_calls, 1)]
    if use_dist:
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = world_kernel_ms[0]
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        world_kernel_ms = [float(v) for v in t.cpu()]
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    n_norm = n_total                                    # what the engine normalises by
    if part_world != world:
        n_total = n_local                               # emulated slab: count what was stacked
    work_step = n_total * ns                            # node-samples per step, whole job
    if time_sharded:
        work_step *= world                              # every rank scanned its own timesteps
    value = work_step * args.steps / elapsed
    kern_s = kern_ms / 1e3 / max(kern_calls, 1)         # avg stacking-kernel time, this rank
    steps_per_launch = eng.get("steps_per_launch") if spl > 1 else 1
    kern_launch_s = kern_s
    kern_s /= steps_per_launch                          # ... per timestep
    # (column partition: the kernel line describes the engine holding the whole planes)
    n_kernel = ((boxes[0][1] - boxes[0][0]) * (boxes[0][3] - boxes[0][2]) * nz
                if by_columns and boxes else n_local)
    local_ns = n_kernel * ns
    b_fused = 4.0 * n_kernel * S + 8.0 * S * t_samples + 24.0 * ns  # SURVEY 8d B_F
    shift_kernel = eng.get("last_kernel") == 3
    # 8-byte LDS operands fetched per add: 1 for the round-2 kernels; the shift-reuse kernel shares
    # a register window between the 8 nodes of a group (measured on the resident table)
    wide_tiles = eng.get("shift_wide_tiles") if shift_kernel else 0
    per_add = (eng.get("shift_wide_operands_per_add_x1000" if wide_tiles else "shift_operands_per_add_x1000") / 1000.0
               if shift_kernel else 1.0)
    exp_ops = EXP_FP64_OPS_LAZY if shift_kernel and eng.get("shift_lazy") == 1 else EXP_FP64_OPS
    chip = onchip(screened, local_ns, S, kern_s, per_add, exp_ops)
    valu_key = "int32_valu" if screened else "fp64_valu"
    # the ceiling that binds: whichever on-chip unit is busier (the fused detect cannot be HBM
    # bound: SURVEY.md section 8d); the HBM-compulsory figure is kept under its own key
    bind = "lds" if chip["lds"]["frac"] >= chip[valu_key]["frac"] else valu_key
    kname = (f"qm::screen_lds_kernel<{eng.get('screen_pairs')}, {(S + 7) // 8}>"
             if screened else stack_kernel_name(eng, S))
    traffic, traffic_detail = (stored_traffic(f"{args.config}:detect", kname, b_fused)
                               if world == 1 and part_world == 1 and spl == 1 else (None, None))
    result = {
        "metric": "grid-nodes x time-samples stacked /sec (detect sweep)",
        "metric_definition": "value = n_nodes x n_samples x steps / wall time of the timed region, the "
                             "travel-time table AND the log-onsets already resident in HBM when the "
                             "clock starts, the three output series left in HBM (the bench contract's "
                             "definition).  SURVEY.md section 8d defines the same metric with the "
                             "per-step H2D of the onsets and D2H of the series inside the timed region: "
                             "that is `step_with_copies.value` (copies overlapped on HIP streams).",
        "value": value, "unit": "node-samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if (args.weak or streaming) else "strong",
        "vs_baseline": None,
        "dtype": "i32 fixed-point sweep + f64 refinement (opt-in screen=1)" if screened else "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.config} detect sweep: grid {grid[0]}x{ny}x{nz}, {S} onset "
                               f"rows, {ns} samples per step, fused migrate+find_max_coa in "
                               f"float64, table resident in HBM, onsets "
                               + ("streamed from pinned host memory (copies inside the timed "
                                  "region)" if streaming else "resident in HBM")
                               + (f"; {x_range[1] - x_range[0]} x-planes of it on rank {part_rank}"
                                  if part_world > 1 else ""),
                   "n_nodes_per_gpu": n_local, "n_rows": S, "n_samples": ns,
                   "steps_per_launch": steps_per_launch,
                   "sharding": ("timesteps round-robin over the ranks, whole grid on every rank"
                                if time_sharded else
                                "flat-index ranges at (x, y)-column granularity (up to 3 boxes per "
                                "rank)" if by_columns else "x-plane slabs" if world > 1 else "none"),
                   "boxes_rank0": [list(b) for b in boxes] if by_columns else None,
                   "exchange": ("1 x all_gather([3 boxes][3][n_samples]) + device fold per step"
                                if by_columns else
                                {"packed": "1 x all_gather([3][n_samples]) + device fold per step",
                                 "allreduce": "3 x all_reduce(n_samples) per step"}[args.exchange]
                                if use_dist and not time_sharded else "none"),
                   "collective_backend": backend, "ranks": world,
                   "ranks_seen": dist.get_world_size() if use_dist else 1,
                   "kernel_ms_per_rank": {"min": min(world_kernel_ms), "max": max(world_kernel_ms)},
                   "engine": dict(tunables, brick=[eng.get("brick_x"), eng.get("brick_y"),
                                                   eng.get("brick_z")],
                                  samples_per_lane=eng.get("samples_per_lane"),
                                  waves=eng.get("waves"))},
        "kernel": {"name": kname, "avg_ms": kern_launch_s * 1e3,
                   "avg_ms_per_step": kern_s * 1e3, "steps_per_launch": steps_per_launch,
                   "launches": kern_calls, "timing": "HIP events on the launch stream",
                   # shift-reuse kernel: how the scan was tiled (wide = 384-sample tiles, six samples per lane,
                   # round 6; behind them 256-sample tiles and / or one tail tile) and on which brick shape
                   "tiles": ({"wide_384": wide_tiles, "tail_samples_per_lane": eng.get("shift_tail_spl"),
                              "samples_per_lane": eng.get("last_kernel_j"),
                              "brick_nodes": eng.get("shift_wide_brick_nodes" if wide_tiles else "shift_brick_nodes"),
                              "lazy_argmax": eng.get("shift_lazy"),
                              "lds_operands_per_add": per_add} if shift_kernel else None)},
        # what the library was built from (generator constants + source digest; "overlay=none; defines=none" =
        # the product build, include/qmhip.h: qm_build_info)
        "build_info": lib.build_info(),
        "roofline": {"bound": bind, "achieved": chip[bind]["achieved"], "peak": chip[bind]["peak"],
                     "unit": chip[bind]["unit"], "frac": chip[bind]["frac"],
                     # HBM bytes per launch (fetch + write) from the stored PMC passes over this
                     # kernel and workload -- null when none is stored for what this run launched
                     "traffic": traffic, "traffic_detail": traffic_detail,
                     "lds_frac": chip["lds"]["frac"], "valu_frac": chip[valu_key]["frac"],
                     # the reference's own arithmetic alone (S float64 adds per node-sample; the
                     # engine's 2^z / sum / arg-max epilogue left out) against the same VALU peak
                     "adds_only_frac": (None if screened else
                                        local_ns * S / kern_s / FP64_PEAK),
                     "hbm_compulsory": {"bound": "hbm", "achieved": b_fused / kern_s / 1e9,
                                        "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                        "frac": b_fused / kern_s / HBM_PEAK,
                                        "algorithmic_bytes_per_launch": b_fused},
                     "note": "fused detect never writes the volume: compulsory HBM bytes are the "
                             "table, the onsets and the outputs only (hbm_compulsory, far below 1 % "
                             "by construction), so the roofline that binds is on the chip: "
                             + (f"FP64 VALU issue (S adds + ~{exp_ops} epilogue instructions per node-"
                                "sample against 39.3e12 instruction-lanes/s at 2.4 GHz), beside "
                                "which the data returning from LDS costs the SIMDs 2 cycles per "
                                "8-byte operand (DESIGN.md section 3.4)" if bind != "lds" else
                                "LDS operand bytes (8 per add against 150 TB/s) with the FP64 "
                                "VALU co-saturated")},
        "roofline_onchip": chip,
    }

    # ---- the same steps fed from the host: H2D + detect + D2H inside the timed region ----
    if world == 1 and not streaming and not args.no_copies:
        from quakemigrate_amd.stream import StreamingDetector

        sd = StreamingDetector(eng, S, t_samples, case.fsmp, case.lsmp, case.available,
                               n_nodes_total=n_norm, depth=3, device=dev, steps_per_launch=spl)
        # set-up + warm.  Round 6 (tools/diag_stream.py with "stream_stamps"): ONCE per process, around the 180th
        # launch of its first stream, the GPU's command processor sits idle for ~35 ms with launches queued --
        # whatever they hold (no copies, no host writes, no events: the same), a pause does not move it, later
        # streams of the process do not see it.  Inside a 400-step window of 0.4-ms steps that one stall read as
        # "x1.27 with copies"; configurations whose launches are that short warm up past it (<= 0.5 s), and the
        # line says how many launches that was.
        warm_launches = 2 if elapsed / args.steps * spl * 256 > 0.5 else 256
        sd.run(host_onsets[i % n_pool] for i in range(warm_launches * spl))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = sd.run(host_onsets[(args.warmup + i) % n_pool] for i in range(args.steps))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        same = all(np.array_equal(a, b.cpu().numpy()) for a, b in zip(got[-1], res))
        result["step_with_copies"] = {
            "ms_per_step": dt * 1e3, "value": work_step / dt, "unit": "node-samples/s",
            "steps": args.steps, "identical_to_resident_run": bool(same),
            "warmup_launches": warm_launches, "input": "pulled by a kernel" if sd.engine.get("stream_pull") != 0 and
            8.0 * S * t_samples * spl <= (1 << 20) else "copy stream",
            "what": "per step: H2D of the log-onsets from pinned host memory "
                    f"({8.0 * S * t_samples / 1e6:.1f} MB) on a copy stream, fused detect, the three "
                    "series written by the launch straight into pinned host memory; the native "
                    "pipeline (qm_stream_*, depth 3) through its thin Python caller; everything inside the "
                    "timed region"}

    # ---- the opt-in screened detect on the same steps, beside the float64 engine --------
    if not screened and world == 1 and not streaming and not args.no_screened:
        sx = lib.Engine(local_rank, **dict(tunables, screen=1))
        sx.set_stream(torch.cuda.current_stream().cuda_stream)
        sx.load_lut(case.traveltimes, node_offset=x_range[0] * ny * nz)
        out_x = tuple(torch.empty_like(o) for o in out)
        n_x = max(2, min(args.steps, 10))
        sx.detect(onsets_dev[last], case.fsmp, case.lsmp, case.available, n_nodes_total=n_norm,
                  out=out_x)
        torch.cuda.synchronize()
        same_idx = bool(torch.equal(out_x[2], res[2]))
        same_coa = bool(torch.equal(out_x[0], res[0]))
        norm_rel = float(((out_x[1] - res[1]).abs() / res[1]).max().item())
        assert same_idx and same_coa and norm_rel < 1e-6, (same_idx, same_coa, norm_rel)
        sx.config("log_timing", 1)
        t0 = time.perf_counter()
        for i in range(n_x):
            sx.detect(onsets_dev[i % n_pool], case.fsmp, case.lsmp, case.available,
                      n_nodes_total=n_norm, out=out_x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_x
        xk_ms, xk_calls = sx.kernel_log()
        xk_s = xk_ms / 1e3 / max(xk_calls, 1)
        result["screened_optin"] = {
            "what": "Engine(screen=1), NOT the default: every node-sample stacked as exact int32 "
                    "fixed point (pairs of samples: ds_read_b64 + v_add3_u32), every (brick, "
                    "sample) cell that can hold the maximum re-evaluated in float64 -- max_coa / "
                    "max_coa_idx are the float64 engine's bits; max_norm_coa within 6.7e-7 of it "
                    "by a deterministic bound whose preconditions are checked per step on the "
                    "device (DESIGN_HISTORY.md section 3.2, qm_screen.hpp)",
            "dtype": "i32 fixed-point sweep + f64 refinement",
            "max_norm_coa_bound": 6.7e-7,
            "ms_per_step": dt * 1e3, "value": work_step / dt, "unit": "node-samples/s",
            "steps": n_x,
            "kernel": {"name": f"qm::screen_lds_kernel<{sx.get('screen_pairs')}, {(S + 7) // 8}>",
                       "avg_ms": xk_s * 1e3, "launches": xk_calls},
            "roofline_onchip": onchip(True, local_ns, S, xk_s),
            "candidate_cells_last_step": sx.get("last_candidates"),
            "steps_screened": sx.get("screened_steps"),
            "steps_fallen_back_to_float64": sx.get("fallback_steps"),
            "vs_float64": {"max_coa_idx_identical": same_idx, "max_coa_identical": same_coa,
                           "max_norm_coa_max_rel_diff": norm_rel}}
        sx.close()

    # ---- the opt-in tie_rule = 1 (the reference's arg-max rule on near-ties) on the same steps ----
    if not screened and world == 1 and not streaming and not args.no_screened and not tunables.get("tie_rule"):
        tx = lib.Engine(local_rank, **dict(tunables, tie_rule=1))
        tx.set_stream(torch.cuda.current_stream().cuda_stream)
        tx.load_lut(case.traveltimes, node_offset=x_range[0] * ny * nz)
        out_t = tuple(torch.empty_like(o) for o in out)
        n_t = max(2, min(args.steps, 10))
        for i in range(2):
            tx.detect(onsets_dev[(last - 1 + i) % n_pool], case.fsmp, case.lsmp, case.available, n_nodes_total=n_norm,
                      out=out_t)
        torch.cuda.synchronize()
        same_idx = bool(torch.equal(out_t[2], res[2]))          # (generic data: no near-ties, the same series)
        same_coa = bool(torch.equal(out_t[0], res[0]))
        norm_rel = float(((out_t[1] - res[1]).abs() / res[1]).max().item())
        tx.config("log_timing", 1)
        t0 = time.perf_counter()
        for i in range(n_t):
            tx.detect(onsets_dev[i % n_pool], case.fsmp, case.lsmp, case.available, n_nodes_total=n_norm, out=out_t)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_t
        tk_ms, tk_calls = tx.kernel_log()
        result["tie_rule_optin"] = {
            "what": "Engine(tie_rule=1), NOT the default: nodes whose sums lie within two ulps of a sample's largest "
                    "are compared on a correctly rounded exp(sum / available), lowest index among equal values -- "
                    "migratelib.c:98-105 as the reference's scalar-libm build computes it (DESIGN.md section 1, "
                    "csrc/qm_ties.hpp).  The stacking kernel also leaves the largest value per brick and sample; "
                    "the refinement re-stacks one brick per sample",
            "ms_per_step": dt * 1e3, "value": work_step / dt, "unit": "node-samples/s", "steps": n_t,
            "vs_default_ms_per_step": dt / (elapsed / args.steps),
            "stacking_kernel_avg_ms": tk_ms / max(tk_calls, 1), "brick_rows": tx.get("tie_brick_rows"),
            "candidate_pairs_last_step": tx.get("tie_pairs"), "overflow_samples": tx.get("tie_overflow_samples"),
            "vs_default": {"max_coa_idx_identical": same_idx, "max_coa_identical": same_coa,
                           "max_norm_coa_max_rel_diff": norm_rel}}
        assert same_idx and same_coa and norm_rel < 1e-12, result["tie_rule_optin"]
        tx.close()

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
        mt, mt_detail = stored_traffic("C3L:volume", stack_kernel_name(eng, S, volume=True), b_mat)
        result["roofline_materialised"] = {
            "kernel": stack_kernel_name(eng, S, volume=True), "traffic": mt, "traffic_detail": mt_detail,
            "bound": "hbm", "achieved": b_mat / sec / 1e9, "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": b_mat / sec / HBM_PEAK, "avg_ms": sec * 1e3,
            "node_samples_per_s": n_local * ns_loc / sec,
            "workload": f"locate window: same grid, {ns_loc} samples, volume "
                        f"({8.0 * n_local * ns_loc / 1e9:.1f} GB) written to HBM + scan"}
        # what this GPU reaches on the same buffer with nothing but stores / loads (context for the
        # fraction above: the volume launch is bound by its arithmetic -- the marginal-map flavour below
        # is the same launch without the stores -- and writes at about half of the pure store rate)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        vol.fill_(1.0)
        e0.record()
        for _ in range(3):
            vol.fill_(1.0)
        e1.record()
        torch.cuda.synchronize()
        fill_s = e0.elapsed_time(e1) / 1e3 / 3
        write_ceiling = 8.0 * n_local * ns_loc / fill_s
        result["roofline_materialised"].update({
            "measured_write_only_GBps": write_ceiling / 1e9,
            "frac_of_measured_write_only": (8.0 * n_local * ns_loc / sec) / write_ceiling,
            "write_only_note": "torch fill_ of the same volume buffer: what a pure store stream reaches on "
                               "this GPU; `frac` above stays against the 8 TB/s HBM figure"})
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

    # ---- a change of station availability: another table takes over (lut.py:529-537) --------------
    if not args.no_table_switch and world == 1 and cfg_name == "C3" and not streaming and spl == 1:
        torch.cuda.synchronize()
        keep = [r for r in range(S) if r not in (3, S - 2)]        # a P and an S row drop out
        tt_b = torch.from_numpy(np.ascontiguousarray(case.traveltimes[..., keep])).to(dev)
        tt_a = torch.from_numpy(case.traveltimes).to(dev)
        on_b = onsets_dev[0][keep].contiguous()
        out_b = tuple(torch.empty_like(o) for o in out)

        def run(which):
            if which == "A":
                eng.detect(onsets_dev[0], case.fsmp, case.lsmp, S, n_nodes_total=n_norm, out=out)
            else:
                eng.detect(on_b, case.fsmp, case.lsmp, S - 2, n_nodes_total=n_norm, out=out_b)

        def timed(fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3

        sw = lib.Engine(local_rank, **tunables)
        sw.set_stream(torch.cuda.current_stream().cuda_stream)
        eng_saved, eng = eng, sw
        # first sight of each table: upload from HBM + brick tables + the kernel's layouts + one step
        first_a = timed(lambda: (sw.select_table("A"), sw.load_lut(tt_a), run("A")))
        first_b = timed(lambda: (sw.select_table("B"), sw.load_lut(tt_b), run("B")))
        assert sw.select_table("A")
        run("A")
        steady_a = min(timed(lambda: run("A")) for _ in range(3))
        # alternating between the two parked tables, one step each
        alt = []
        for i in range(6):
            which = "B" if i % 2 == 0 else "A"
            alt.append(timed(lambda: (sw.select_table(which), run(which))))
        assert sw.get("table_misses") == 2
        same = bool(torch.equal(out[2], res[2])) if last == 0 else None
        result["table_switch"] = {
            "what": "station availability changes: table A = the 30 rows, table B = 28 of them; first "
                    "sight of a table = load from HBM + derived layouts + one step; afterwards "
                    "Engine.select_table swaps the parked device state in (no device work) and the "
                    "step runs at its usual time",
            "first_step_with_new_table_ms": {"A": first_a, "B": first_b},
            "steady_step_ms": steady_a,
            "alternating_step_ms": alt,
            "table_switch_ms": max(0.0, float(np.mean(alt[2:])) - steady_a),
            "rebuild_ms": first_b - steady_a,
            "device_bytes_per_table": sw.get("table_bytes"),
            "parked_bytes": sw.get("tables_parked_bytes"),
            "result_unchanged": same}
        eng = eng_saved
        sw.close()
        del tt_a, tt_b

    # ---- BASELINE configs[2] literally: the whole n_samples volume of the step written to HBM with its scan
    # (SURVEY 8d figure 1 on the configuration itself: 196 GB at C3).  Part of the default line since round 6
    # (VERDICT r05 item 3), skipped -- and said so -- where the GPU has not the memory free.
    full_bytes = 8.0 * n_local * ns
    free_bytes = torch.cuda.mem_get_info(dev)[0] if world == 1 else 0
    want_full = (args.materialise_full or (cfg_name == "C3" and not args.no_materialise_full)) and \
        world == 1 and not streaming and part_world == 1
    if want_full and free_bytes < full_bytes + (4 << 30):
        lib.release_cached_memory() if hasattr(lib, "release_cached_memory") else None
        torch.cuda.empty_cache()
        free_bytes = torch.cuda.mem_get_info(dev)[0]
    if want_full and free_bytes < full_bytes + (4 << 30):
        result["roofline_materialised_full"] = {"skipped": f"{full_bytes / 1e9:.0f} GB volume, "
                                                           f"{free_bytes / 1e9:.0f} GB of HBM free"}
    elif want_full:
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
        k_full = stack_kernel_name(eng, S, volume=True)
        tiles_full = {"wide_384": eng.get("shift_wide_tiles"), "tail_samples_per_lane": eng.get("shift_tail_spl")}
        eng.detect(onsets_dev[0], case.fsmp, case.lsmp, case.available, n_nodes_total=n_total,
                   out=out)
        torch.cuda.synchronize()
        result["roofline_materialised_full"] = {
            "kernel": k_full, "tiles": tiles_full,
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
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(result) + "\n").encode())
    os.close(result_fd)


if __name__ == "__main__":
    main()
