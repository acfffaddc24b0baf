#!/usr/bin/env bash
# (ran at commit 689b07b, whose kernels_batch_gemm.hip has the experimental options; reverted afterwards — results: profiles/r03_loader_kernel_experiments.txt)
# round 3, lease H: where the loader-wave kernels' time goes — the 64-slot step with one operand's DMA / the MFMAs / the epilogue left
# out (gemv_probe bits; wrong results), per-kernel times under rocprofv3
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
for pr in 0 1 2 3 4 8 7 15; do
  DTK_OPTIONS="gemv_probe=$pr" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_h" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 12 --fork > "$OUT/prof_h.log" 2>&1
  db=$(ls "$OUT"/prof_h/*/*.db "$OUT"/prof_h/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r03h_probe_$pr.csv" > /dev/null
  rm -rf "$OUT/prof_h"
  echo "== gemv_probe=$pr  $(grep ms/step $OUT/prof_h.log)"; grep -E "k_gemv_bl|k_gemv_bkl" "$OUT/r03h_probe_$pr.csv" | cut -c1-110
done
