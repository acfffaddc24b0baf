#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -s -k "batch or parallel" -p no:cacheprovider > "$OUT/pytest_batch.log" 2>&1
echo "pytest(batch) exit $?"; tail -8 "$OUT/pytest_batch.log"
timeout 300 python tools/probe_batch.py 2>&1 | tail -2
for b in 1 8 16; do timeout 300 python tools/bench_batch.py --batch $b --steps 48 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_batch" -o trace -- python "$REPO/tools/bench_batch.py" --batch 8 --steps 32 > "$OUT/prof_batch.log" 2>&1
cd "$REPO"; python tools/prof_summary.py "$OUT/prof_batch/trace_results.db" "$OUT/batch_kernel_stats.csv" > "$OUT/batch_stats.txt" 2>&1; head -9 "$OUT/batch_stats.txt"
