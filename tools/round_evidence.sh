#!/bin/bash
# A round's evidence run on a GPU box (via gpurun), in parts so that a call can be kept short:
#   suite    the GPU suite + smoke()
#   bench    every bench line: C3 headline (with cpu_baseline, step_with_copies, table_switch, locate rooflines),
#            its round-2 / round-3 forms, C2, C1 / E1 / E2 / E2F with 1 and 8 timesteps per launch, a C4 slab,
#            the C5 stream and the literal 720-step day; tables of 66 / 128 / 200 rows (row blocks)
#   ranks    N > 1 plumbing on one GPU (two gloo ranks; a one-rank RCCL group), the enqueue budget of a sharded step
#   stats    rocprofv3 --kernel-trace --stats of a bench command whose only launches of the headline kernel are
#            the warm-up and timed steps (its average must agree with the bench line's own HIP-event clock)
#   pmc      separate rocprofv3 --pmc passes (SQ, GRBM, FETCH_SIZE, WRITE_SIZE) over the detect, locate-volume,
#            marginal-map launches at C3 and the detect launch on a C4 slab; the traffic file bench.py reads
#   widen    the rows SURVEY 8(f) marks next + the drop-in migrate's volume rate
#   tie      the opt-in tie_rule = 1 beside the default engine (C3 / C2 / C1; round 5's form of it: tie_sets = 0), the
#            kernel stats of a C3 step with the rule on
#   stall    one timestep per launch on example-sized grids: the GPU-clock stamps around every launch of a process's
#            first stream (the one-off stall), the rate behind it, later streams of the same process
# usage: tools/round_evidence.sh <tag> [part ...]      (default: all parts)     outputs: gpurun_out/<tag>/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:?tag}; shift
PARTS=${*:-suite bench ranks stats pmc widen tie stall}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
has() { [[ " $PARTS " == *" $1 "* ]]; }
line() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d.get("step_with_copies") or {}
        print(f.split("/")[-1], "ms/step", round(d["ms_per_step"], 4), "value %.4e" % d["value"], d["kernel"]["name"],
              "frac", round(d["roofline"]["frac"], 3), d["roofline"]["bound"],
              ("| with copies %.4f (x%.3f)" % (c["ms_per_step"], c["ms_per_step"] / d["ms_per_step"])) if c else "")
    except Exception as e:
        print(f, "unreadable:", e)
PY
}
if has suite; then
  ( time python -m pytest tests -q -m gpu --durations=6 ) > $OUT/pytest_gpu.log 2>&1; tail -12 $OUT/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
fi
if has bench; then
  python bench.py --steps 20 --warmup 3 > $OUT/bench_C3.json 2> $OUT/bench.err; tail -3 $OUT/bench.err
  Q="--no-cpu-baseline --no-screened --no-copies --no-table-switch"
  python bench.py --steps 20 --warmup 3 --engine '{"shift_wide": 0}' $Q --no-materialised > $OUT/bench_C3_round5_tiles.json 2>> $OUT/bench.err
  python bench.py --steps 20 --warmup 3 --engine '{"shift_wide": 0, "shift_tail": 0}' $Q --no-materialised > $OUT/bench_C3_round3_tiles.json 2>> $OUT/bench.err
  python bench.py --steps 20 --warmup 3 --engine '{"shift": 0}' $Q --no-materialised > $OUT/bench_C3_round2_kernels.json 2>> $OUT/bench.err
  python bench.py --config C2 --steps 30 --warmup 3 --no-cpu-baseline --no-materialised > $OUT/bench_C2.json 2>> $OUT/bench.err
  for cfg in C1 E1 E2 E2F; do for k in 1 8; do
    python bench.py --config $cfg --steps 400 --warmup 8 --steps-per-launch $k --no-cpu-baseline --no-materialised --no-screened --no-table-switch > $OUT/bench_${cfg}_k$k.json 2>> $OUT/bench.err
  done; done
  python bench.py --config C4 --emulate-world 8 --emulate-rank 3 --steps 5 --warmup 1 --no-cpu-baseline --no-materialised > $OUT/bench_C4_slab3of8.json 2>> $OUT/bench.err
  python bench.py --config C5 --steps 30 --warmup 3 > $OUT/bench_C5_stream.json 2>> $OUT/bench.err
  python bench.py --config C5 --steps 720 --warmup 3 > $OUT/bench_C5_24h.json 2>> $OUT/bench.err
  line $OUT/bench_*.json | tee $OUT/bench_lines.txt
  # tables of more than 64 rows (row blocks) on the C3 grid x 1536 samples, and a 401-sample volume on one
  { for r in 66 128 200; do python tools/ab.py --config C3 --case "{\"rows\": $r, \"n_samples\": 1536}" - | sed "s/^-/rows $r/"; done
    python tools/ab.py --config C3 --mode volume --case '{"rows": 128, "n_samples": 401}' - | sed "s/^-/rows 128 volume/"; } 2>&1 | grep -v amdgpu.ids | tee $OUT/row_blocks.txt
fi
if has ranks; then
  # N > 1 plumbing on one GPU (gloo rendezvous; RCCL refuses two ranks on one device): self-launch
  QM_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 1 > $OUT/bench_C3_2ranks_one_gpu.json 2> $OUT/bench_2ranks.err; tail -c 300 $OUT/bench_C3_2ranks_one_gpu.json; tail -2 $OUT/bench_2ranks.err
  QM_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --config C5 --steps 6 --warmup 1 > $OUT/bench_C5_2ranks_one_gpu.json 2>> $OUT/bench_2ranks.err
  QM_BENCH_FORCE_DIST=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-screened --no-copies --no-materialised --no-table-switch > $OUT/bench_C3_one_rank_rccl.json 2>> $OUT/bench_2ranks.err; tail -c 300 $OUT/bench_C3_one_rank_rccl.json
  python tools/enqueue_budget.py --world 8 --rank 3 > $OUT/enqueue_C3_rank3of8.json 2> $OUT/enqueue.err; cat $OUT/enqueue_C3_rank3of8.json
fi
if has stats; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- \
      python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-screened --no-table-switch --no-copies --no-materialised > $OUT/bench_C3_headline_only_under_rocprof.json 2> $OUT/prof.err
  find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/bench_C3_headline_only_kernel_stats.csv \;
  head -6 $OUT/bench_C3_headline_only_kernel_stats.csv
  python -c "
import json; d=json.loads(open('$OUT/bench_C3_headline_only_under_rocprof.json').read().strip().splitlines()[-1]); print('bench line under rocprof: kernel avg_ms', d['kernel']['avg_ms'], 'launches', d['kernel']['launches'], 'ms/step', d['ms_per_step'])"
  find $OUT/prof -name "*.csv" -size +1M -delete
  # (round 6) the same kind of line for the other kernels beside their HIP-event times: the
  # locate window's volume and marginal-map launches, a C4 slab's detect, a 128-row table's row-block kernel
  stat() {  # tag, tune.py arguments...
    local tag=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o $tag -- \
        python $ROOT/tools/tune.py --reps 6 --sweep '[{}]' "$@" > $OUT/stats_$tag.log 2>&1
    find $OUT/prof_$tag -name "*kernel_stats.csv" -exec cp {} $OUT/${tag}_kernel_stats.csv \;
    grep -E "stack_shift|Name" $OUT/${tag}_kernel_stats.csv | head -3; grep -E "ms|kernel" $OUT/stats_$tag.log | tail -2
    find $OUT/prof_$tag -name "*.csv" -size +1M -delete
  }
  stat C3L_volume --config C3 --ns 401 --volume
  stat C3L_marginal --config C3 --ns 401 --marginal
  stat C4_slab --config C4 --x-range 150,200
  stat rows128 --config C3 --rows 128 --ns 1536
  cd $ROOT
fi
if has pmc; then
  bash tools/prof_counters.sh C3 '[{}]' $TAG/pmc_shift > $OUT/pmc_shift.txt 2>&1; grep -E "stack_" $OUT/pmc_shift.txt | head
  bash tools/prof_counters.sh C3 '[{}]' $TAG/pmc_locate "--ns 401 --volume" > $OUT/pmc_locate.txt 2>&1; grep -E "stack_" $OUT/pmc_locate.txt | head
  bash tools/prof_counters.sh C3 '[{}]' $TAG/pmc_marginal "--ns 401 --marginal" > $OUT/pmc_marginal.txt 2>&1; grep -E "stack_" $OUT/pmc_marginal.txt | head
  bash tools/prof_counters.sh C4 '[{}]' $TAG/pmc_C4slab "--x-range 150,200 --ns 3072" > $OUT/pmc_C4slab.txt 2>&1; grep -E "stack_" $OUT/pmc_C4slab.txt | head
  f() { find $OUT/$1/$2 -name "*counter_collection.csv" | head -1; }
  python tools/pmc_traffic.py "C3:detect=$(f pmc_shift fetch),$(f pmc_shift write)" "C3L:volume=$(f pmc_locate fetch),$(f pmc_locate write)" "C3L:marginal=$(f pmc_marginal fetch),$(f pmc_marginal write)" > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err; head -40 $OUT/pmc_traffic.json
fi
if has widen; then
  python tools/widen_bench.py > $OUT/widen_rows.jsonl 2> $OUT/widen.err; cat $OUT/widen_rows.jsonl
fi
if has tie; then
  for c in C3 C2 C1; do
    python tools/ab.py --config $c --steps 8 --engines '[{}, {"tie_rule": 1}, {"tie_rule": 1, "tie_sets": 0}]' - 2>&1 | grep -v amdgpu.ids
  done | tee $OUT/tie_ab.txt
  ( cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tie -o tie -- \
        python $ROOT/tools/tune.py --config C3 --reps 6 --sweep '[{"tie_rule": 1}]' > $OUT/stats_tie.log 2>&1
    find $OUT/prof_tie -name "*kernel_stats.csv" -exec cp {} $OUT/C3_tie_rule_kernel_stats.csv \;
    find $OUT/prof_tie -name "*.csv" -size +1M -delete )
  grep -E "stack_shift|tie_|fill_|combine|Name" $OUT/C3_tie_rule_kernel_stats.csv | cut -c1-150
fi
if has stall; then
  { echo "# C1, one timestep per launch, depth 3: GPU-clock stamps around every launch of the process's first stream"
    python tools/diag_stream.py C1 1 3 600 '{"stream_stamps": 1}' 2>&1 | grep -v amdgpu.ids
    echo "# the same with the copy stream instead of the pull kernel"
    python tools/diag_stream.py C1 1 3 600 '{"stream_stamps": 1, "stream_pull": 0}' 2>&1 | grep -v amdgpu.ids
    echo "# a pause of 0.3 s before the clock starts does not move it"
    DIAG_SLEEP=0.3 python tools/diag_stream.py C1 1 3 600 2>&1 | grep -v amdgpu.ids
    echo "# 300 launches before the clock starts: the rate behind it (C1, E2, E1 at K = 1; C1 at K = 8)"
    for c in C1 E2 E1; do DIAG_WARM=300 python tools/diag_stream.py $c 1 3 600 '{"stream_stamps": 1}' 2>&1 | grep -v amdgpu.ids; done
    DIAG_WARM=40 python tools/diag_stream.py C1 8 3 800 2>&1 | grep -v amdgpu.ids
    echo "# three streams one after the other in one process, then detect + device synchronisation per step"
    python tools/diag_stall.py 2>&1 | grep -v amdgpu.ids; } | cut -c1-420 | tee $OUT/stream_stall.txt
fi
