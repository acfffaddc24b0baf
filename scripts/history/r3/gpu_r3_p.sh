#!/usr/bin/env bash
# round 3, lease P: the pair + V tile qkv kernel with two loader waves and a ring of 5 phases (four in flight), now that the compute
# waves keep their accumulators in VGPRs: identity tests, per-kernel times
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "x_once_per_cu" -p no:cacheprovider 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
run() {
  local opt=$1; shift
  DTK_OPTIONS="$opt" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_p" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 12 --fork "$@" > "$OUT/prof_p.log" 2>&1
  db=$(ls "$OUT"/prof_p/*/*.db "$OUT"/prof_p/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r03p_tmp.csv" > /dev/null
  rm -rf "$OUT/prof_p"
  echo "== $opt $*  $(grep ms/step $OUT/prof_p.log)"; grep -E "k_gemv_bl<2|k_gemv_b<2" "$OUT/r03p_tmp.csv" | cut -c1-120
}
run "gemv_bl=1"
run "gemv_bl=9,gemv_loaders=1"
run "gemv_bl=9,gemv_loaders=2"
run "gemv_bl=9,gemv_loaders=2,gemv_xw=1"
run "gemv_bl=13,gemv_loaders=2" --model detikzify-cl-7b --weight-format fp8
