#!/usr/bin/env bash
# round 6, lease K2 — lease K's step times again with the option set explicitly on both sides (dtk_set_option's kernel switches are process-wide:
# lease K's "" runs inherited gemv_bks=0 from the run before them), and the x-bandwidth probe with its slice rounding fixed.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
R=${R:-r06k2}
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
SB=$REPO/tools/probe/step_bench
{
echo "== cl-7b fp8, 64 slots, 8 images"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=48 timeout 300 $SB "gemv_bks=0" "gemv_bks=1" "gemv_bks=0" "gemv_bks=1"
echo "== ds-7b bf16, 64 slots"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=48 timeout 300 $SB "gemv_bks=0" "gemv_bks=1" "gemv_bks=0" "gemv_bks=1"
echo "== ds-1.3b bf16, 64 slots"
STEP_BENCH_MODEL=ds-1.3b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=48 timeout 300 $SB "gemv_bks=0" "gemv_bks=1" "gemv_bks=0" "gemv_bks=1"
} 2>&1 | sed -E 's/; last token.*//' | tee "$OUT/${R}_step_bench.txt"
timeout 120 tools/probe/xbw_probe 2>&1 | tee "$OUT/${R}_xbw_probe.txt"
