#!/usr/bin/env bash
# round 6, lease N2 — lease N's A/B with 200-step runs, three alternations (48-step runs scatter by 3-4 % on some boxes).
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
R=${R:-r06n2}
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
SB=$REPO/tools/probe/step_bench
A="gemv_bus=0,gemv_bks=0"; B="gemv_bus=0,gemv_bks=1"; C="gemv_bus=1,gemv_bks=1"; D="gemv_bus=3,gemv_bks=1"
{
echo "== cl-7b fp8, 64 slots, 8 images, 200 steps (context 247 .. 447)"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=200 timeout 600 $SB "$A" "$B" "$C" "$D" "$A" "$B" "$C" "$D" "$A" "$B" "$C" "$D"
echo "== ds-7b bf16, 64 slots, 200 steps"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=200 timeout 600 $SB "$A" "$B" "$C" "$D" "$A" "$B" "$C" "$D" "$A" "$B" "$C" "$D"
} 2>&1 | sed -E 's/; last token.*//' | tee "$OUT/${R}_step_bench.txt"
