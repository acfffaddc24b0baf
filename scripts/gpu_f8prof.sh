#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
for f in bf16 fp8; do timeout 300 python tools/bench_batch.py --model detikzify-cl-7b --batch 16 --steps 48 --weight-format $f 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_f8b" -o trace -- python "$REPO/tools/bench_batch.py" --model detikzify-cl-7b --batch 16 --steps 32 --weight-format fp8 > "$OUT/prof_f8b.log" 2>&1
echo "rocprof exit $?"
cd "$REPO"; python tools/prof_summary.py "$OUT/prof_f8b/trace_results.db" "$OUT/f8b_kernel_stats.csv" | head -14
