#!/usr/bin/env bash
# round 6, lease L — k_gemv_bus (qkv / gate-up at 64 slots: a block per CU, its 8 waves = the 8 K slices, operands straight into registers): logits hash and
# step times per role against the kernels it replaces (every kernel switch set explicitly: they are process-wide), then the variants test.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06l}
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
SB=$REPO/tools/probe/step_bench
V='"gemv_bus=0" "gemv_bus=1" "gemv_bus=2" "gemv_bus=3" "gemv_bus=0" "gemv_bus=3"'
{
echo "== cl-7b fp8, 64 slots, 8 images"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=48 timeout 300 $SB "gemv_bus=0" "gemv_bus=1" "gemv_bus=2" "gemv_bus=3" "gemv_bus=0" "gemv_bus=3"
echo "== ds-7b bf16, 64 slots"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=48 timeout 300 $SB "gemv_bus=0" "gemv_bus=1" "gemv_bus=2" "gemv_bus=3" "gemv_bus=0" "gemv_bus=3"
echo "== ds-1.3b bf16, 64 slots"
STEP_BENCH_MODEL=ds-1.3b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=48 timeout 300 $SB "gemv_bus=0" "gemv_bus=1" "gemv_bus=2" "gemv_bus=3" "gemv_bus=0" "gemv_bus=3"
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
  rm -rf "$OUT/prof_$name" "$OUT/prof_$name.log"; echo "-- $name (try $try)"; head -9 "$OUT/${R}_$name.csv" | cut -c1-150
}
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=24 prof batch64_fp8_kernel_stats "" $SB "gemv_bus=3"
STEP_BENCH_SLOTS=64 STEP_BENCH_MODEL=ds-7b STEP_BENCH_STEPS=24 prof batch64_kernel_stats "" $SB "gemv_bus=3"
cd "$REPO"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "x_once_per_cu" 2>&1 | tail -5 | tee "$OUT/${R}_pytest.txt"
