#!/usr/bin/env bash
# round 2, call P: per-kernel totals over the whole batched phase of bench.py (64 rollouts x 512 tokens, growing contexts)
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_p" -o trace -- python "$REPO/bench.py" --steps 1 --warmup 0 --mcts-trees 0 --no-cpu-baseline --probe-tokens 2 > "$OUT/prof_p.log" 2>&1; echo "rocprof exit $?"
python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_p -name trace_results.db | head -1)" "$OUT/r02_bench_batched_phase_kernel_stats.csv" > /dev/null 2>&1
rm -rf "$OUT/prof_p"; head -22 "$OUT/r02_bench_batched_phase_kernel_stats.csv" | cut -c1-170
