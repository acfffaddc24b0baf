#!/usr/bin/env bash
# round 2, call C: per-kernel profiles (rocprofv3 kernel-trace) of the single-sequence step and of the batched step under three option sets
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
prof() {  # name, env options, command...
  local name=$1 opts=$2; shift 2
  DTK_OPTIONS="$opts" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
  echo "prof $name exit $?"
  python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_$name -name trace_results.db | head -1)" "$OUT/r02_${name}_kernel_stats.csv" > /dev/null 2>&1
  rm -rf "$OUT/prof_$name"
  head -14 "$OUT/r02_${name}_kernel_stats.csv" | cut -c1-150
}
prof decode "" python "$REPO/bench.py" --steps 1 --warmup 0 --new-tokens 128 --no-cpu-baseline --batch 0 --mcts-seq-expansions 0 --probe-tokens 4
prof batch_tail "attn_b_impl=1,prefix_mfma=0,tail_threads=256,gemv_b_wide=0" python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork
prof batch_prefix "attn_b_impl=1,prefix_mfma=1,pfx_splits=4,tail_threads=256,gemv_b_wide=0" python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork
prof batch_wide "attn_b_impl=1,prefix_mfma=0,tail_threads=256,gemv_b_wide=1" python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork
grep "ms/step" $OUT/prof_batch_*.log
