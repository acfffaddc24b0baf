#!/usr/bin/env bash
# round 6, lease N — k_gemv_bus with its epilogue operands requested before the k loop; k_gemv_bks back on 16-k-step chunks for fp8; variants test.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06n}
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
SB=$REPO/tools/probe/step_bench
{
echo "== cl-7b fp8, 64 slots, 8 images"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=48 timeout 300 $SB "gemv_bus=0,gemv_bks=0" "gemv_bus=0,gemv_bks=1" "gemv_bus=1,gemv_bks=1" "gemv_bus=2,gemv_bks=1" "gemv_bus=3,gemv_bks=1" "gemv_bus=0,gemv_bks=0" "gemv_bus=3,gemv_bks=1"
echo "== ds-7b bf16, 64 slots"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=48 timeout 300 $SB "gemv_bus=0,gemv_bks=0" "gemv_bus=1,gemv_bks=1" "gemv_bus=3,gemv_bks=1" "gemv_bus=0,gemv_bks=0" "gemv_bus=1,gemv_bks=1"
} 2>&1 | sed -E 's/; last token.*//' | tee "$OUT/${R}_step_bench.txt"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "x_once_per_cu" 2>&1 | tail -5 | tee "$OUT/${R}_pytest.txt"
