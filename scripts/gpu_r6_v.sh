#!/usr/bin/env bash
# round 6, lease V — what bounds a k_gemm_g3 block at M = 243: the kernel with its fills left out (1), its MFMAs left out (2), both (3)
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06v}
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift 1
  for try in 1 2 3; do
    rm -rf "$OUT/prof_$name"
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
    local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/${R}_$name.csv" > /dev/null && break
  done
  rm -rf "$OUT/prof_$name" "$OUT/prof_$name.log"; echo "-- $name (try $try)"
  grep -i "gemm_g3\|sk_reduce" "$OUT/${R}_$name.csv" | cut -c1-170
}
for pr in 0 1 2 3; do DTK_G3_PROBE=$pr prof probe$pr python "$REPO/tools/bench_prefill.py" --sets "prefill_sk=1" --reps 3 --rows 16; done 2>&1 | tee "$OUT/${R}_g3_probe.txt"
