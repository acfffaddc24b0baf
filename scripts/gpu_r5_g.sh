#!/usr/bin/env bash
# round 5, lease G: lease F's config 4 (16 slots, 513 steps) waited 3.80 ms per step against round 4's 3.29 — the 2-wave attention
# blocks that win at 64 slots lose at 16 once the private context is long.  Block size x prefix kernel at 16 / 32 / 64 slots and
# 4 / 260 / 480 private keys, ds-7b bf16.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
SB=$REPO/tools/probe/step_bench
{
for slots in 16 32 64; do for warm in 4 260 480; do
  echo "== ds-7b bf16, $slots slots, 1 image, $warm private keys"
  STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=$slots STEP_BENCH_WARM=$warm STEP_BENCH_STEPS=24 timeout 200 $SB "" "tail_threads=256" "tail_threads=512" "prefix_mfma=0,tail_threads=256"
done; done
} 2>&1 | sed -E 's/; logits hash.*//' | tee "$OUT/r05g_step_bench.txt"
