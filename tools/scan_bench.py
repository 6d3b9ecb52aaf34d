# -*- coding: utf-8 -*-
"""find_max_coa on a device-resident volume: HBM-read roofline of scan_volume_kernel (dev aid)."""
import sys, pathlib, json, time
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from quakemigrate_amd.core import lib

def main():
    eng = lib.Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for n_nodes, ns in [(4080501, 401), (520251, 6000), (259008, 625)]:
        vol = torch.rand((n_nodes, ns), dtype=torch.float64, device="cuda") + 0.5
        out = tuple(torch.empty(ns, dtype=d, device="cuda") for d in (torch.float64, torch.float64, torch.int64))
        eng.find_max_coa(vol, ns, n_nodes, out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            eng.find_max_coa(vol, ns, n_nodes, out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        ref_idx = torch.argmax(vol, dim=0)
        ok = bool(torch.equal(ref_idx, out[2]))
        print(json.dumps(dict(n_nodes=n_nodes, ns=ns, ms=round(ms, 3), GBps=round(8.0 * n_nodes * ns / ms / 1e6, 1),
                              frac_of_8TBps=round(8.0 * n_nodes * ns / ms / 1e6 / 8000, 3), argmax_ok=ok)))
        del vol

if __name__ == "__main__":
    main()
