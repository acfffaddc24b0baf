#!/usr/bin/env bash
# round 6, lease E: k_gemv_bc, second version (cached flags, the DONE words in one LDS read, the epilogue's operands requested before
# the k loop): the 64-slot step off / on / forced units, per-kernel times, bit-identity test.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
SB=$REPO/tools/probe/step_bench
{
echo "== cl-7b fp8, 64 slots, 8 images, 4 private keys"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=32 timeout 600 $SB "gemv_bc=0" "" "gemv_bc=23" "gemv_bc=39" "gemv_bc=55"
echo "== ds-7b bf16, 64 slots, 1 image, 4 private keys"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=32 timeout 600 $SB "gemv_bc=0" "" "gemv_bc=39" "gemv_bc=55"
echo "== ds-1.3b bf16, 64 slots"
STEP_BENCH_MODEL=ds-1.3b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=32 timeout 600 $SB "gemv_bc=0" ""
} 2>&1 | sed -E 's/; last token.*//' | tee "$OUT/r06e_step_bench.txt"
cd /tmp && export TMPDIR=/tmp
prof() {   # name, counters ("" = kernel stats), command...
  local name=$1 ctrs=$2; shift 2
  if [ -z "$ctrs" ]; then timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
  else timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/prof_$name" -o pmc -- "$@" > "$OUT/prof_$name.log" 2>&1; fi
  local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r06e_$name.csv" $([ -n "$ctrs" ] && echo --pmc) > /dev/null
  rm -rf "$OUT/prof_$name" "$OUT/prof_$name.log"; echo "-- $name"; head -8 "$OUT/r06e_$name.csv" | cut -c1-150
}
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=24 prof batch64_fp8_kernel_stats "" $SB ""
STEP_BENCH_SLOTS=64 STEP_BENCH_MODEL=ds-7b STEP_BENCH_STEPS=24 prof batch64_kernel_stats "" $SB ""
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q -k "x_once_per_cu" 2>&1 | tail -4
