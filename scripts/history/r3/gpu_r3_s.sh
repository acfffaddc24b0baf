#!/usr/bin/env bash
# (ran at commit ef278c3, which has k_gemv_br; removed afterwards — results: profiles/r03_loader_kernel_experiments.txt)
# round 3, lease S: k_gemv_br (weights through registers with a hand-counted 4-phase ring, x through a 6-phase LDS-DMA ring):
# identity tests, per-kernel times of the 64-slot step
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "x_once_per_cu" -p no:cacheprovider 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
run() {
  local opt=$1; shift
  DTK_OPTIONS="$opt" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_s" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 12 --fork "$@" > "$OUT/prof_s.log" 2>&1
  db=$(ls "$OUT"/prof_s/*/*.db "$OUT"/prof_s/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r03s_tmp.csv" > /dev/null
  rm -rf "$OUT/prof_s"
  echo "== $opt $*  $(grep ms/step $OUT/prof_s.log)"; grep -E "k_gemv_bl|k_gemv_br|k_gemv_b<2|k_gemv_bx" "$OUT/r03s_tmp.csv" | cut -c1-120
}
run "gemv_bl=1"
run "gemv_bl=33"
run "gemv_bl=35"
run "gemv_bl=39" --model detikzify-cl-7b --weight-format fp8
