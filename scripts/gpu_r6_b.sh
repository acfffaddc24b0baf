#!/usr/bin/env bash
# round 6, lease B: k_gemv_bc (a compute wave per column tile) — bit-identity against k_gemv_b and its twins, then the 64-slot step with
# it off / on / forced units per block (fp8 config-5 shape and ds-7b bf16), per-kernel times under rocprofv3; the real-checkpoint
# procedure on the synthetic checkpoints; the tightened smoke().
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
SB=$REPO/tools/probe/step_bench
{
echo "== cl-7b fp8, 64 slots, 8 images, 4 private keys"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=32 timeout 600 $SB "gemv_bc=0" "" "gemv_bc=3" "gemv_bc=1" "gemv_bc=23" "gemv_bc=39" "gemv_bc=55"
echo "== cl-7b fp8, 64 slots, 8 images, 260 private keys"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_WARM=260 STEP_BENCH_STEPS=32 timeout 600 $SB "gemv_bc=0" ""
echo "== ds-7b bf16, 64 slots, 1 image, 4 private keys"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=32 timeout 600 $SB "gemv_bc=0" "" "gemv_bc=1" "gemv_bc=2" "gemv_bc=4"
echo "== ds-1.3b bf16, 64 slots"
STEP_BENCH_MODEL=ds-1.3b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=32 timeout 600 $SB "gemv_bc=0" ""
} 2>&1 | tee "$OUT/r06b_step_bench.txt"
cd /tmp && export TMPDIR=/tmp
prof() {   # name, counters ("" = kernel stats), command...
  local name=$1 ctrs=$2; shift 2
  if [ -z "$ctrs" ]; then timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
  else timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/prof_$name" -o pmc -- "$@" > "$OUT/prof_$name.log" 2>&1; fi
  local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r06b_$name.csv" $([ -n "$ctrs" ] && echo --pmc) > /dev/null
  rm -rf "$OUT/prof_$name"; echo "-- $name"; head -9 "$OUT/r06b_$name.csv" | cut -c1-150
}
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 prof batch64_fp8_kernel_stats "" $SB ""
STEP_BENCH_SLOTS=64 STEP_BENCH_MODEL=ds-7b prof batch64_kernel_stats "" $SB ""
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 prof batch64_fp8_pmc_fetch "FETCH_SIZE" $SB ""
cd $REPO
timeout 1500 python -m pytest tests -m gpu -x -q -k "x_once_per_cu or real_checkpoint or 32_slot_batch_matches or lds_staged" 2>&1 | tail -15 | tee "$OUT/r06b_pytest.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/r06b_smoke.txt"
