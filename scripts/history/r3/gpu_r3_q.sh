#!/usr/bin/env bash
# round 3, lease Q: TWO x waves (ordinary loads, two phases of lead) so that the LDS-DMA loader issues weight pieces only: identity
# tests, per-kernel times of the 64-slot step
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x_once_per_cu" -p no:cacheprovider 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
run() {
  local opt=$1; shift
  DTK_OPTIONS="$opt" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_q" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 12 --fork "$@" > "$OUT/prof_q.log" 2>&1
  db=$(ls "$OUT"/prof_q/*/*.db "$OUT"/prof_q/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r03q_tmp.csv" > /dev/null
  rm -rf "$OUT/prof_q"
  echo "== $opt $*  $(grep ms/step $OUT/prof_q.log)"; grep -E "k_gemv_bl|k_gemv_bkl|k_gemv_b<2|k_gemv_bx" "$OUT/r03q_tmp.csv" | cut -c1-120
}
run "gemv_xw=0"
run "gemv_xw=2"
run "gemv_xw=2,gemv_bl=3"
run "gemv_xw=2,gemv_bl=9"
run "gemv_xw=2,gemv_bl=7" --model detikzify-cl-7b --weight-format fp8
