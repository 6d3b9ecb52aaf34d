#!/bin/bash
# copy the summaries of a tools/round_evidence.sh run (gpurun_out/<tag>/) into profiles/ under a round prefix
# usage: tools/collect_evidence.sh <tag> <round prefix, e.g. r05>
cd "$(dirname "$0")/.."
O=gpurun_out/${1:?tag}; R=${2:?round prefix}
for f in $O/bench_*.json $O/enqueue_*.json; do [ -f "$f" ] && cp $f profiles/${R}_$(basename $f); done
[ -f $O/bench_lines.txt ] && cp $O/bench_lines.txt profiles/${R}_bench_lines.txt
[ -f $O/row_blocks.txt ] && cp $O/row_blocks.txt profiles/${R}_row_blocks.txt
[ -f $O/pytest_gpu.log ] && { tail -14 $O/pytest_gpu.log; tail -1 $O/smoke.log; } > profiles/${R}_pytest_gpu_tail.txt
[ -f $O/bench_C3_headline_only_kernel_stats.csv ] && cp $O/bench_C3_headline_only_kernel_stats.csv profiles/${R}_bench_C3_headline_only_kernel_stats.csv
[ -f $O/pmc_traffic.json ] && cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json
# (round 6) --kernel-trace --stats lines of the other kernels: only the stacking kernels' rows
for t in C3L_volume C3L_marginal C4_slab rows128; do
  [ -f $O/${t}_kernel_stats.csv ] && { head -1 $O/${t}_kernel_stats.csv; grep -E "stack_shift" $O/${t}_kernel_stats.csv; } > profiles/${R}_${t}_kernel_stats.csv
done
[ -f $O/widen_rows.jsonl ] && cp $O/widen_rows.jsonl profiles/${R}_widen_rows.jsonl
# (round 6, second half) the opt-in tie_rule = 1 beside the default engine; the stream pipeline's one-off stall
[ -f $O/tie_ab.txt ] && cp $O/tie_ab.txt profiles/${R}_tie_ab.txt
[ -f $O/C3_tie_rule_kernel_stats.csv ] && { head -1 $O/C3_tie_rule_kernel_stats.csv; grep -E "stack_shift|tie_|fill_|combine" $O/C3_tie_rule_kernel_stats.csv; } > profiles/${R}_C3_tie_rule_kernel_stats.csv
[ -f $O/stream_stall.txt ] && cp $O/stream_stall.txt profiles/${R}_stream_stall.txt
for pair in shift:C3shift locate:C3locate marginal:C3marginal C4slab:C4slab; do
  k=${pair%%:*}; n=${pair##*:}
  [ -d $O/pmc_$k ] || continue
  for c in sq1 sq2 grbm fetch write; do
    f=$(find $O/pmc_$k/$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f profiles/${R}_pmc_${n}_$c.csv
  done
  cp $O/pmc_$k.txt profiles/${R}_pmc_${n}_summary.txt
done
ls profiles | grep -c "^${R}_"
