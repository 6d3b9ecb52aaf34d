# Time the C3 slab each rank of an N-GPU strong-scaled run would hold, one at a time on one GPU
# (development aid for the x-plane partition: shows what slab shapes cost beside 1/N of the grid).
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/${1:-slabs}; mkdir -p $out
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-screened 2>/dev/null | tail -1 > $out/full.json
for w in 2 4 8; do
  for r in 0 $((w/2)) $((w-1)); do
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-screened --emulate-world $w --emulate-rank $r 2>/dev/null | tail -1 > $out/w${w}_r${r}.json
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d = json.loads(open(f).read())
        print(f.split("/")[-1], d["ms_per_step"], d["value"], d["config"])
    except Exception as ex:
        print(f, "unreadable", ex)
PY
