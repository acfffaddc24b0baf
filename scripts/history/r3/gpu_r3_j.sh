#!/usr/bin/env bash
# round 3, lease J: x fragments by an extra wave (ordinary loads + ds_write_b128) instead of LDS-DMA pieces, accumulators kept in VGPRs:
# identity tests, per-kernel times of the 64-slot step per option set
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x_once_per_cu" -p no:cacheprovider 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
run() {  # model-args..., then DTK_OPTIONS as $1
  local opt=$1; shift
  DTK_OPTIONS="$opt" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_j" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 12 --fork "$@" > "$OUT/prof_j.log" 2>&1
  db=$(ls "$OUT"/prof_j/*/*.db "$OUT"/prof_j/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r03j_tmp.csv" > /dev/null
  rm -rf "$OUT/prof_j"
  echo "== $opt $*  $(grep ms/step $OUT/prof_j.log)"; grep -E "k_gemv_bl|k_gemv_bkl|k_gemv_b<|k_gemv_bx" "$OUT/r03j_tmp.csv" | cut -c1-110
}
run "gemv_xw=0"
run "gemv_xw=1"
run "gemv_xw=1,gemv_bl=9"
run "gemv_xw=1,gemv_bl=3"
run "gemv_xw=0,gemv_bl=9"
run "gemv_xw=0" --model detikzify-cl-7b --weight-format fp8
run "gemv_xw=1,gemv_bl=5" --model detikzify-cl-7b --weight-format fp8
run "gemv_xw=1,gemv_bl=13" --model detikzify-cl-7b --weight-format fp8
