#!/usr/bin/env bash
# round 6, lease D: what bounds k_gemv_bc — parts left out (option gemv_bc_probe: bit 0 no x loads, 1 no weight DMA, 2 no LDS reads /
# MFMAs, 3 no epilogue; wrong results, timing only), fp8 config-5 shape and ds-7b bf16, 64 slots.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
SB=$REPO/tools/probe/step_bench
{
echo "== cl-7b fp8, 64 slots, 8 images"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=32 timeout 600 $SB "" "gemv_bc_probe=8" "gemv_bc_probe=9" "gemv_bc_probe=10" "gemv_bc_probe=12" "gemv_bc_probe=11" "gemv_bc_probe=13" "gemv_bc_probe=14" "gemv_bc_probe=15" "gemv_bc=0"
echo "== ds-7b bf16, 64 slots"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=32 timeout 600 $SB "" "gemv_bc_probe=8" "gemv_bc_probe=9" "gemv_bc_probe=10" "gemv_bc_probe=12" "gemv_bc_probe=11" "gemv_bc_probe=14" "gemv_bc_probe=15" "gemv_bc=0"
} 2>&1 | sed -E 's/; logits hash.*//' | tee "$OUT/r06d_bc_probe.txt"
