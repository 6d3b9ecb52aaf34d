#!/bin/bash
# The first run on a real multi-GPU MI355X node, in the order that isolates a failure fastest.
# Nothing here has been run on more than one GPU (no such node was available to any round); what one
# GPU can show -- the N > 1 code path over gloo, the RCCL calls on a one-rank group, the host's
# enqueue cost per sharded step (tools/enqueue_budget.py: 0.10 ms of a 5.8 ms step) -- is recorded
# under profiles/.  usage: tools/preflight_8gpu.sh [N=8] [out-dir]
set -u
N=${1:-8}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${2:-$ROOT/gpurun_out/preflight_${N}gpu}
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0            # dmabuf IPC: without it RCCL fails with hipIpcGetMemHandle
export MASTER_ADDR=127.0.0.1
fail() { echo "PREFLIGHT FAILED at: $*"; exit 1; }
check() { # json file, expected ranks
  python - "$1" "$2" <<'PY' || exit 1
import json, sys
d = json.load(open(sys.argv[1]))
c = d["config"]
want = int(sys.argv[2])
k = c["kernel_ms_per_rank"]
print(f"  {sys.argv[1].split('/')[-1]}: value {d['value']:.4e} {d['unit']}, {d['ms_per_step']:.3f} ms/step, "
      f"ranks_seen {c['ranks_seen']}, backend {c['collective_backend']}, "
      f"kernel ms per rank min {k['min']:.3f} max {k['max']:.3f}, sharding: {c['sharding']}")
assert c["ranks_seen"] == want, f"ranks_seen {c['ranks_seen']} != {want}"
assert c["collective_backend"] == "nccl", c["collective_backend"]
assert k["max"] <= 1.15 * k["min"] + 0.05, "ranks are unbalanced by more than 15 %"
PY
}
echo "== 0. devices"; rocm-smi --showid 2>/dev/null | grep -c "GPU\[" ; python -c "import torch; n = torch.cuda.device_count(); print('torch sees', n, 'GPUs'); assert n >= $N" || fail "fewer than $N GPUs visible"
echo "== 1. one GPU: suite + smoke (the code is sound on this node's GPU 0)"
python -m pytest tests -q -m gpu -x > "$OUT/pytest_gpu.log" 2>&1 || fail "pytest -m gpu (see $OUT/pytest_gpu.log)"
tail -1 "$OUT/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke()" || fail "smoke"
echo "== 2. N = 1 bench line (the reference point of the scaling curve)"
python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_C3_n1.json" 2> "$OUT/bench_n1.err" || fail "bench N=1"
check "$OUT/bench_C3_n1.json" 1
for n in 2 4 $N; do
  [ "$n" -gt "$N" ] && continue
  echo "== 3. C3 sharded over $n GPUs (strong scaling: the same grid, column partition, ONE packed all-gather per step)"
  python bench.py --gpus $n --config C3 --steps 20 --warmup 3 > "$OUT/bench_C3_n$n.json" 2> "$OUT/bench_C3_n$n.err" || fail "bench C3 N=$n (see $OUT/bench_C3_n$n.err)"
  check "$OUT/bench_C3_n$n.json" $n
done
echo "== 4. C4 (401x401x201 x 60 rows x 12000 samples) over $N GPUs -- BASELINE configs[3]"
python bench.py --gpus $N --config C4 --steps 5 --warmup 1 > "$OUT/bench_C4_n$N.json" 2> "$OUT/bench_C4_n$N.err" || fail "bench C4 N=$N"
check "$OUT/bench_C4_n$N.json" $N
echo "== 5. C5: the 24 h stream, timesteps round-robin over the ranks, no collective in the data path"
python bench.py --gpus $N --config C5 --steps 90 --warmup 3 > "$OUT/bench_C5_n$N.json" 2> "$OUT/bench_C5_n$N.err" || fail "bench C5 N=$N"
check "$OUT/bench_C5_n$N.json" $N
echo "== 6. the three-all-reduce form of the exchange and the plane partition give the same series"
python bench.py --gpus $N --exchange allreduce --partition planes --steps 5 --warmup 1 > "$OUT/bench_C3_n${N}_allreduce.json" 2> "$OUT/bench_allreduce.err" || fail "bench allreduce"
check "$OUT/bench_C3_n${N}_allreduce.json" $N
echo "expected at N = 8 on C3 (from the one-GPU slab timings, DESIGN.md section 5): kernel ~5.9 ms per rank, "
echo "step ~6.0-6.3 ms with the exchange (one 432 KB all-gather + a fold of 24 sets), i.e. ~7.2-7.5x of N = 1."
echo "PREFLIGHT OK -> $OUT"
