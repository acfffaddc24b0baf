#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python tools/bench_vit.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee "$OUT/r06ah_vit_tiles.txt"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof_v"; timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_v" -o trace -- python "$REPO/tools/bench_vit.py" --only 1 > "$OUT/prof_v.log" 2>&1
db=$(ls "$OUT"/prof_v/*/*.db "$OUT"/prof_v/*.db 2>/dev/null | head -1); [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r06ah_vit1_kernel_stats.csv" > /dev/null; rm -rf "$OUT/prof_v" "$OUT/prof_v.log"
head -12 "$OUT/r06ah_vit1_kernel_stats.csv" | cut -c1-150
