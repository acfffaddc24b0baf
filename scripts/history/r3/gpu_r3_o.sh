#!/usr/bin/env bash
# round 3, lease O: per-kernel totals over the whole batched phase of bench.py (64 rollouts x 512 tokens, growing contexts) — the
# per-layer breakdown over a rollout that DESIGN §3.1b quotes; ds-7b and cl-7b fp8
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
for cfg in "detikzify-ds-7b bf16 ds7b" "detikzify-cl-7b fp8 cl7b_fp8"; do
  set -- $cfg
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_o" -o trace -- python "$REPO/bench.py" --model $1 --weight-format $2 --steps 1 --warmup 0 --mcts-trees 0 --mcts-seq-expansions 0 --no-config4 --no-config5 --no-cpu-baseline --probe-tokens 2 > "$OUT/prof_o_$3.log" 2>&1; echo "rocprof exit $?"
  python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_o -name trace_results.db | head -1)" "$OUT/r03_bench_batched_phase_$3_kernel_stats.csv" > /dev/null 2>&1
  rm -rf "$OUT/prof_o"; head -12 "$OUT/r03_bench_batched_phase_$3_kernel_stats.csv" | cut -c1-170
  grep -o '"rollouts_per_sec": [0-9.]*' "$OUT/prof_o_$3.log" | head -3
done
