#!/usr/bin/env bash
# (ran at commit 689b07b, whose kernels_batch_gemm.hip has the experimental options; reverted afterwards — results: profiles/r03_loader_kernel_experiments.txt)
# round 3, lease I: the loader-wave kernels with all LDS fragment reads of a phase issued before its MFMAs: identity tests, then
# per-kernel times of the 64-slot step for (loaders, qkv form, probe) combinations
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x_once_per_cu" -p no:cacheprovider 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for opt in "gemv_probe=0" "gemv_probe=2" "gemv_probe=4" "gemv_bl=9" "gemv_bl=3" "gemv_loaders=2,gemv_bl=9" "gemv_loaders=2,gemv_bl=9,gemv_probe=4" "gemv_loaders=2,gemv_bl=9,gemv_probe=2" "gemv_bl=9,gemv_probe=4" "gemv_bl=9,gemv_probe=5" "gemv_bl=9,gemv_probe=6"; do
  DTK_OPTIONS="$opt" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_i" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 12 --fork > "$OUT/prof_i.log" 2>&1
  db=$(ls "$OUT"/prof_i/*/*.db "$OUT"/prof_i/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r03i_tmp.csv" > /dev/null
  rm -rf "$OUT/prof_i"
  echo "== $opt  $(grep ms/step $OUT/prof_i.log)"; grep -E "k_gemv_bl|k_gemv_bkl|k_gemv_b<" "$OUT/r03i_tmp.csv" | cut -c1-110
done
