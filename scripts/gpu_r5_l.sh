#!/usr/bin/env bash
# round 5, lease L: last option sweep of the 64-slot attention (tail block 64 / 128 threads, 2 / 3 / 4 prefix key splits) at 4 and 260
# private keys, ds-7b bf16 and cl-7b fp8 with 8 images.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
SB=$REPO/tools/probe/step_bench
{
for warm in 4 260; do
  echo "== ds-7b bf16, 64 slots, 1 image, $warm private keys"
  STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_WARM=$warm STEP_BENCH_STEPS=32 timeout 300 $SB "" "tail_threads=64" "pfx_splits=2" "pfx_splits=3" ""
  echo "== cl-7b fp8, 64 slots, 8 images, $warm private keys"
  STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_WARM=$warm STEP_BENCH_STEPS=32 timeout 300 $SB "" "tail_threads=64" "pfx_splits=2" "pfx_splits=3"
done
} 2>&1 | sed -E 's/; logits hash.*//' | tee "$OUT/r05l_step_bench.txt"
