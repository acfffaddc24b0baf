#!/usr/bin/env bash
# round 6, lease I: after the prune (DTK_EXPERIMENTS families, dtk_decode_batch_run, image_prep) and the looped prefix groups: the GPU
# tests those touch; the 64-slot step again; where a ds-1.3b token goes (kernel trace of the single-sequence step: item 7's ceiling).
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
gcc -O2 -Iinclude examples/c_abi_smoke.c -o build/c_abi_smoke -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" && ./build/c_abi_smoke 2>&1 | tail -3
SB=$REPO/tools/probe/step_bench
{
echo "== cl-7b fp8, 64 slots, 8 images (lease F, 16 groups: 3.67 ms; lease H, 64 grid rows: 3.73-3.86)"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=48 timeout 600 $SB "" "" "gemv_bc=0"
echo "== ds-7b bf16, 64 slots, 1 image (lease F: 3.95-4.04; lease H: 4.03-4.10)"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=48 timeout 600 $SB "" ""
echo "== 16 slots (BASELINE config 4's context), ds-7b"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=16 STEP_BENCH_STEPS=48 timeout 600 $SB ""
} 2>&1 | sed -E 's/; last token.*//' | tee "$OUT/r06i_step_bench.txt"
cd /tmp && export TMPDIR=/tmp
prof() {   # name, counters ("" = kernel stats), command...
  local name=$1 ctrs=$2; shift 2
  if [ -z "$ctrs" ]; then timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
  else timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/prof_$name" -o pmc -- "$@" > "$OUT/prof_$name.log" 2>&1; fi
  local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r06_$name.csv" $([ -n "$ctrs" ] && echo --pmc) > /dev/null
  rm -rf "$OUT/prof_$name" "$OUT/prof_$name.log"; echo "-- $name"; head -10 "$OUT/r06_$name.csv" | cut -c1-150
}
STEP_BENCH_MODEL=ds-1.3b STEP_BENCH_SLOTS=1 STEP_BENCH_STEPS=256 prof ds13b_single_kernel_stats "" $SB ""
cd $REPO
timeout 1500 python -m pytest tests -m gpu -x -q -k "attn or x_once_per_cu or 32_slot or gemm or engine or kv_fork or smoke or abi or real_checkpoint" 2>&1 | tail -6 | tee "$OUT/r06i_pytest.txt"
