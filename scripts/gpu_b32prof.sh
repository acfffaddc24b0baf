#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_b32" -o trace -- python "$REPO/tools/bench_batch.py" --model detikzify-cl-7b --batch 32 --steps 32 > "$OUT/prof_b32.log" 2>&1
echo "rocprof exit $?"
cd "$REPO"; python tools/prof_summary.py "$OUT/prof_b32/trace_results.db" "$OUT/b32_kernel_stats.csv" | grep -v "gemm_mfma\|fill\|quant\|retile\|rows" | head -9
