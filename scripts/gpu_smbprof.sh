#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_smb" -o trace -- python "$REPO/bench.py" --model detikzify-v2-8b --sample --steps 1 --warmup 0 --new-tokens 64 --no-cpu-baseline --probe-tokens 2 --batch 0 > "$OUT/prof_smb.log" 2>&1
echo "rocprof exit $?"
cd "$REPO"; python tools/prof_summary.py "$OUT/prof_smb/trace_results.db" "$OUT/smb_kernel_stats.csv" | grep -i "smb\|sample" | cut -c1-200
