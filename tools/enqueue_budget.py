#!/usr/bin/env python3
"""
Host-side cost of one step of an N-way sharded detect, measured on ONE GPU (development aid; the
first real multi-GPU run should hold no surprises of this kind).

Rank `--rank` of a `--world`-way column partition of C3 (or C4) holds up to three boxes, i.e. up
to three engines: per step the host enqueues three stacking launches + three combines, ONE
collective (here on a one-rank RCCL group: the same RCCL calls, nothing to wait for) and the fold.
The kernels of such a rank take ~6 ms at 8-way C3; what the HOST needs per step must stay well
below that or the GPU idles between steps.  Printed: the host time of each call of a step
(enqueue only, the stream is not waited for), the step's enqueue total, and the step time with
the GPU in the loop (enqueue far ahead of execution: the stream's own time per step).

usage (GPU box): python tools/enqueue_budget.py [--config C3] [--world 8] [--rank 3] [--steps 60]
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--partition", default="columns", choices=["columns", "planes"])
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    from quakemigrate_amd import distributed as qd
    from quakemigrate_amd import synth
    from quakemigrate_amd.core import lib

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)                                       # (RCCL's banner goes to the C-level stdout)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    nx, ny, nz = synth.CONFIGS[args.config]["grid"]
    if args.partition == "columns":
        boxes = qd.column_boxes(*qd.shard_columns(nx, ny, args.world, args.rank), ny)
    else:
        x0, x1 = qd.shard_planes(nx, args.world, args.rank)
        boxes = [(x0, x1, 0, ny)]
    engines, case = [], None
    for (bx0, bx1, by0, by1) in boxes:
        c = synth.make_case(args.config, step=0, x_range=(bx0, bx1))
        case = case or c
        eng = lib.Engine(0)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.load_lut(np.ascontiguousarray(c.traveltimes[:, by0:by1]), node_offset=(bx0 * ny + by0) * nz)
        engines.append(eng)
    ns = case.n_samples
    lon = torch.from_numpy(np.log(np.clip(case.onsets, 0.01, np.inf))).to(dev)
    if args.partition == "columns":
        det = qd.ColumnShardedDetector(engines, nx * ny * nz, ns, dev, fold_engine=engines[0])
    else:
        det = qd.ShardedDetector(engines[0], nx * ny * nz, ns, dev)
    for _ in range(3):
        det.detect(lon, case.fsmp, case.lsmp, case.available)
    torch.cuda.synchronize()

    # (a) the stream's time per step: enqueue runs ahead, one synchronize at the end
    t0 = time.perf_counter()
    for _ in range(args.steps):
        det.detect(lon, case.fsmp, case.lsmp, case.available)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0

    # (b) per call, each step drained first so that no call waits for queue space
    calls = {}

    def timed(name, fn):
        t = time.perf_counter()
        out = fn()
        calls.setdefault(name, []).append(time.perf_counter() - t)
        return out

    for _ in range(20):
        torch.cuda.synchronize()
        if args.partition == "columns":
            det._bind_stream()
            for k, eng in enumerate(det.engines):
                timed(f"detect_partial[{k}]", lambda: eng.detect_partial(
                    lon, case.fsmp, case.lsmp, case.available,
                    (det.packed[k, 0], det.packed[k, 1].view(torch.int64), det.packed[k, 2])))
            timed("all_gather", lambda: qd.all_gather_packed(det.packed, det.gathered, det.group))
            timed("fold", lambda: det.fold_engine.finalize_packed(
                det.gathered, det.world * qd.MAX_BOXES, ns, det.n_nodes_total, out=det.out))
        else:
            timed("detect", lambda: det.detect(lon, case.fsmp, case.lsmp, case.available))
    torch.cuda.synchronize()
    per_call = {k: float(np.median(v)) * 1e3 for k, v in calls.items()}
    kern = []
    for eng in engines:
        eng.config("log_timing", 1)
    for _ in range(5):
        det.detect(lon, case.fsmp, case.lsmp, case.available)
    torch.cuda.synchronize()
    for eng in engines:
        ms, n = eng.kernel_log()
        kern.append(ms / max(n, 1))
    result = {
        "config": args.config, "world": args.world, "rank": args.rank, "partition": args.partition,
        "boxes": [list(b) for b in boxes], "steps": args.steps,
        "stacking_kernel_ms_per_box": kern, "stacking_kernel_ms_per_step": float(sum(kern)),
        "step_ms_gpu_in_the_loop": t_all / args.steps * 1e3,
        "host_loop_ms_per_step_while_enqueueing": t_enq / args.steps * 1e3,
        "host_ms_per_call_drained": per_call,
        "host_enqueue_ms_per_step_drained": float(sum(per_call.values())),
        "enqueue_share_of_step": float(sum(per_call.values())) / (t_all / args.steps * 1e3),
    }
    dist.destroy_process_group()
    os.write(saved, (json.dumps(result) + "\n").encode())


if __name__ == "__main__":
    main()
