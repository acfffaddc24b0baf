#!/usr/bin/env bash
# (ran at commit 689b07b, whose kernels_batch_gemm.hip has the experimental options; reverted afterwards — results: profiles/r03_loader_kernel_experiments.txt)
# round 3, lease F: qkv as a RoPE pair unit + a V row tile per block (gemv_bl bit 3), k_gemv_bkl with two row tiles per compute wave
# (gemv_bkl 2): identity tests, 64-slot step time per variant, per-kernel times of the best
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x_once_per_cu" -p no:cacheprovider 2>&1 | tail -3
for opt in "gemv_bl=1" "gemv_bl=9" "gemv_bl=1,gemv_bkl=2" "gemv_bl=9,gemv_bkl=2"; do
  echo "== $opt"; DTK_OPTIONS=$opt timeout 300 python tools/bench_batch.py --batch 64 --steps 96 --fork 2>&1 | tail -1
done
for opt in "gemv_bl=0" "gemv_bl=12"; do
  echo "== cl-7b fp8 $opt"; DTK_OPTIONS=$opt timeout 300 python tools/bench_batch.py --model detikzify-cl-7b --weight-format fp8 --batch 64 --steps 96 --fork 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp
DTK_OPTIONS="gemv_bl=9,gemv_bkl=2" timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_f" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork > "$OUT/prof_f.log" 2>&1
db=$(ls "$OUT"/prof_f/*/*.db "$OUT"/prof_f/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r03f_batch64_q3_bkl2_kernel_stats.csv" > /dev/null
rm -rf "$OUT/prof_f"; head -9 "$OUT/r03f_batch64_q3_bkl2_kernel_stats.csv" | cut -c1-150
