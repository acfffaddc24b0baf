#!/usr/bin/env bash
# round 6, lease O — after k_gemv_bks and the timing probes were removed: the variants test (now with the GQA model), the batched oracle tests,
# 200-step A/B of k_gemv_bus per model, 64-slot kernel traces + FETCH_SIZE with it.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06o}
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
SB=$REPO/tools/probe/step_bench
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "x_once_per_cu" 2>&1 | tail -5 | tee "$OUT/${R}_pytest.txt"
{
echo "== cl-7b fp8, 64 slots, 8 images, 200 steps (context 247 .. 447): k_gemv_bus off / default"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=200 timeout 600 $SB "gemv_bus=0" "gemv_bus=128" "gemv_bus=0" "gemv_bus=128"
echo "== ds-7b bf16, 64 slots, 200 steps"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=200 timeout 600 $SB "gemv_bus=0" "gemv_bus=128" "gemv_bus=0" "gemv_bus=128"
echo "== the 48-step figures of the earlier leases (context 247 .. 295)"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=48 timeout 600 $SB "gemv_bus=0" "gemv_bus=128"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=48 timeout 600 $SB "gemv_bus=0" "gemv_bus=128"
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
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=24 prof batch64_fp8_kernel_stats "" $SB ""
STEP_BENCH_SLOTS=64 STEP_BENCH_MODEL=ds-7b STEP_BENCH_STEPS=24 prof batch64_kernel_stats "" $SB ""
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=24 prof batch64_fp8_pmc_fetch "FETCH_SIZE" $SB ""
STEP_BENCH_SLOTS=64 STEP_BENCH_MODEL=ds-7b STEP_BENCH_STEPS=24 prof batch64_pmc_fetch "FETCH_SIZE" $SB ""
cd "$REPO"
timeout 1800 python -m pytest tests/test_gpu_parity_batched.py -m gpu -q -x 2>&1 | tail -6 | tee -a "$OUT/${R}_pytest.txt"
