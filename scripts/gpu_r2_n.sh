#!/usr/bin/env bash
# round 2, call N: what the batched attention kernel costs as a function of context (shared prefix / private keys) and block size
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
: > "$OUT/r2n_attn_sweep.txt"
for tt in 64 128 256; do
for cfg in "8 --fork" "250 --fork" "500 --fork" "250 --private" "500 --private"; do
  set -- $cfg
  DTK_OPTIONS="tail_threads=$tt" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_n" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 12 --ctx $1 $2 > "$OUT/prof_n.log" 2>&1
  python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_n -name trace_results.db | head -1)" "$OUT/prof_n.csv" > /dev/null 2>&1
  echo "tail_threads $tt ctx $1 $2: $(grep k_attn_tail_b "$OUT/prof_n.csv" | cut -d, -f1-6)" | tee -a "$OUT/r2n_attn_sweep.txt"
  rm -rf "$OUT/prof_n"
done; done
