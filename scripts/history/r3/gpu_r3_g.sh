#!/usr/bin/env bash
# (ran at commit 689b07b, whose kernels_batch_gemm.hip has the experimental options; reverted afterwards — results: profiles/r03_loader_kernel_experiments.txt)
# round 3, lease G: TWO loader waves per block (alternate phases; one wave's vmcnt holds 63 pieces = 63 KiB in flight): identity tests,
# 64-slot step time per variant, per-kernel times
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x_once_per_cu" -p no:cacheprovider 2>&1 | tail -3
for opt in "gemv_loaders=1" "gemv_loaders=2" "gemv_loaders=2,gemv_bl=9" "gemv_loaders=2,gemv_bl=9,gemv_bkl=2"; do
  echo "== $opt"; DTK_OPTIONS=$opt timeout 300 python tools/bench_batch.py --batch 64 --steps 96 --fork 2>&1 | tail -1
done
for opt in "gemv_loaders=1" "gemv_loaders=2,gemv_bl=5" "gemv_loaders=2,gemv_bl=13"; do
  echo "== cl-7b fp8 $opt"; DTK_OPTIONS=$opt timeout 300 python tools/bench_batch.py --model detikzify-cl-7b --weight-format fp8 --batch 64 --steps 96 --fork 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp
DTK_OPTIONS="gemv_loaders=2,gemv_bl=9" timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_g" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork > "$OUT/prof_g.log" 2>&1
db=$(ls "$OUT"/prof_g/*/*.db "$OUT"/prof_g/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r03g_batch64_two_loaders_kernel_stats.csv" > /dev/null
rm -rf "$OUT/prof_g"; head -9 "$OUT/r03g_batch64_two_loaders_kernel_stats.csv" | cut -c1-150
