#!/usr/bin/env bash
# round 6, lease M — k_gemv_bks with fp8 weights over 32-k-step chunks (a deeper weight ring), k_gemv_bus ring depth 2 / 3 / 4 on qkv.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
R=${R:-r06m}
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
SB=$REPO/tools/probe/step_bench
{
echo "== cl-7b fp8, 64 slots, 8 images: bks off / on, bus off"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=48 timeout 300 $SB "gemv_bus=0,gemv_bks=0" "gemv_bus=0,gemv_bks=1" "gemv_bus=0,gemv_bks=0" "gemv_bus=0,gemv_bks=1" "gemv_bus=3,gemv_bks=1"
for d in 12 0 14; do echo "== qkv by k_gemv_bus, DTK_BUS_PROBE=$d (10 + ring depth; 0 = 3)"; DTK_BUS_PROBE=$d STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=48 timeout 300 $SB "gemv_bus=1,gemv_bks=1" "gemv_bus=1,gemv_bks=1"; done
} 2>&1 | sed -E 's/; last token.*//' | tee "$OUT/${R}_step_bench.txt"
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1 ctrs=$2; shift 2
  for try in 1 2 3; do
    rm -rf "$OUT/prof_$name"
    if [ -z "$ctrs" ]; then timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
    else timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/prof_$name" -o pmc -- "$@" > "$OUT/prof_$name.log" 2>&1; fi
    local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/${R}_$name.csv" $([ -n "$ctrs" ] && echo --pmc) > /dev/null && break
  done
  rm -rf "$OUT/prof_$name" "$OUT/prof_$name.log"; echo "-- $name (try $try)"; head -12 "$OUT/${R}_$name.csv" | cut -c1-150
}
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=24 prof batch64_fp8_kernel_stats "" $SB "gemv_bus=3"
