#!/usr/bin/env bash
# round 4, lease F: the 64-slot unit kernels (k_gemv_bx / _bl / _br) with every epilogue load issued up front — identity tests,
# step times bf16 / fp8, kernel traces; then the two new full-size tests that have not run yet (sampled peaked set, v2-8b at 64 slots).
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "x_once_per_cu or 32_slot_batch or fp8_weights or batched_decode_tracks" 2>&1 | tail -4
for args in "" "--model detikzify-cl-7b --weight-format fp8"; do echo "-- 64 slots $args: $(timeout 300 python tools/bench_batch.py --batch 64 --fork --steps 32 $args 2>&1 | tail -1)"; done
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
  local db; db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r04_${name}_kernel_stats.csv" > /dev/null
  rm -rf "$OUT/prof_$name"
  echo "== $name: $(grep ms/step "$OUT/prof_$name.log")"; grep -E "gemv|attn|norm" "$OUT/r04_${name}_kernel_stats.csv" | head -8 | cut -c1-150
}
prof batch64_bf16_hoisted python "$REPO/tools/bench_batch.py" --batch 64 --fork --steps 16
prof batch64_fp8_hoisted python "$REPO/tools/bench_batch.py" --batch 64 --fork --steps 16 --model detikzify-cl-7b --weight-format fp8
cd "$REPO"
timeout 1500 python -m pytest tests/test_gpu_parity_batched.py -q -p no:cacheprovider -s --durations=4 -k "peaked or v2_8b" 2>&1 | grep -E "^peaked|^batched|passed|failed|Error|assert|^[0-9.]+s " | cut -c1-1600
