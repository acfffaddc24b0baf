#!/usr/bin/env bash
# round 6, lease F: which roles k_gemv_bc takes by default — qkv / gate-up / lm_head on and off, fp8 (config 5's shape) and bf16, 64 slots,
# at the prefix context and at 260 private keys.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
SB=$REPO/tools/probe/step_bench
{
echo "== cl-7b fp8, 64 slots, 8 images, 4 private keys"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=48 timeout 600 $SB "gemv_bc=0" "gemv_bc=1" "gemv_bc=5" "gemv_bc=4" "gemv_bc=7" "gemv_bc=0" "gemv_bc=1"
echo "== ds-7b bf16, 64 slots, 1 image, 4 private keys"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=48 timeout 600 $SB "gemv_bc=0" "gemv_bc=1" "gemv_bc=3" "gemv_bc=7" "gemv_bc=4" "gemv_bc=0" "gemv_bc=3"
echo "== ds-1.3b bf16, 64 slots"
STEP_BENCH_MODEL=ds-1.3b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=48 timeout 600 $SB "gemv_bc=0" "gemv_bc=1" "gemv_bc=3" "gemv_bc=7"
} 2>&1 | sed -E 's/; last token.*//' | tee "$OUT/r06f_step_bench.txt"
