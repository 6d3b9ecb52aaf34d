#!/bin/bash
# round 5, first GPU call: packed records with ONE 64-bit shift per node pair (A/B), the C4 slab's
# shift layout, tile costs of the locate window (64 .. 512 samples) for the cover DP
cd "$(dirname "$0")/.."
O=gpurun_out/r05_check1; mkdir -p $O
V=build_variants
{
python tools/ab.py --config C3 --steps 8 - $V/libqmhip_packed64.so - $V/libqmhip_packed64.so
python tools/ab.py --config C3 --steps 4 --case '{"rows": 128, "n_samples": 1536}' - $V/libqmhip_old32.so - $V/libqmhip_old32.so
python tools/ab.py --config C3L --mode marginal --steps 8 - $V/libqmhip_packed64.so
python tools/ab.py --config C3L --mode volume --steps 8 - $V/libqmhip_packed64.so
python tools/ab.py --config C1 --steps 20 - $V/libqmhip_packed64.so
python tools/ab.py --config C4 --steps 3 --case '{"x_range": [150, 200]}' - $V/libqmhip_packed64.so
} 2>&1 | tee $O/ab.txt
python - <<'PY' 2>&1 | tee $O/tiles.txt
import json, numpy as np, torch, sys
sys.path.insert(0, ".")
from quakemigrate_amd import synth
from quakemigrate_amd.core import lib
case = synth.make_case("C4", x_range=[150, 200], n_samples=512)
eng = lib.Engine(0); eng.load_lut(case.traveltimes)
lon = np.log(np.clip(case.onsets, 0.01, np.inf))
eng.detect(lon, case.fsmp, case.lsmp, case.available)
print("C4 slab:", {k: eng.get(k) for k in ("shift_ok", "shift_waves", "shift_brick_nodes", "shift_wide_bricks",
      "shift_operands_per_add_x1000", "shift_lazy", "n_bricks")})
eng.close()
# tile costs at the C3 grid: marginal map / volume / detect over ns = 64 .. 768
base = synth.make_case("C3L")
eng = lib.Engine(0); eng.load_lut(base.traveltimes)
n = int(np.prod(base.traveltimes.shape[:3]))
for ns in (64, 128, 192, 256, 320, 384, 401, 448, 512, 576, 640, 768):
    c = synth.make_case("C3L", n_samples=ns, table=False)
    lon = torch.from_numpy(np.log(np.clip(c.onsets, 0.01, np.inf))).cuda()
    cmap = torch.zeros(n, dtype=torch.float64, device="cuda")
    out = (torch.zeros(ns, dtype=torch.float64, device="cuda"), torch.zeros(ns, dtype=torch.float64, device="cuda"),
           torch.zeros(ns, dtype=torch.int64, device="cuda"))
    res = {"ns": ns}
    for mode in ("marginal", "detect"):
        best = 1e9
        for _ in range(5):
            if mode == "marginal":
                eng.marginal_map(lon, c.fsmp, c.lsmp, c.available, 0, ns, out=cmap, scan_out=out)
            else:
                eng.detect(lon, c.fsmp, c.lsmp, c.available, out=out)
            best = min(best, eng.last_kernel_ms())
        res[mode] = round(best, 3)
    res["tail_spl"] = eng.get("shift_tail_spl")
    print(json.dumps(res), flush=True)
PY
