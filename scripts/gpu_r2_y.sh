#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_y" -o trace -- python "$REPO/tools/bench_prefill.py" > "$OUT/prof_y.log" 2>&1
python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_y -name trace_results.db | head -1)" "$OUT/prof_y.csv" > /dev/null 2>&1
rm -rf "$OUT/prof_y"; grep -E "k_gemm|k_retile|k_attention|rmsnorm|silu|rope" "$OUT/prof_y.csv" | cut -c1-150
