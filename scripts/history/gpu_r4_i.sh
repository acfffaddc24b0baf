#!/usr/bin/env bash
# round 4, lease I: the fp8 matrix-core step with the operand layout the probe measured (lease H: the quantiser tests passed, every
# GEMV was 30 % off — ck_tile's "32 consecutive k per lane" is not what the scaled instruction does; profiles/r04_mx_probe.txt):
# op-level + two-layer tests, step times, kernel trace, the full-depth cl-7b fp8 test, and the two tests fixed after lease G.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity_mx.py -q -p no:cacheprovider -s --tb=short 2>&1 | grep -vE "amdgpu.ids|^$" | cut -c1-1200 | tail -120 | tee "$OUT/r04i_mx_tests.txt"
for b in 64 32 16; do
  echo "-- cl-7b fp8, $b slots, act_fp8=1: $(timeout 300 python tools/bench_batch.py --batch $b --fork --steps 32 --model detikzify-cl-7b --weight-format fp8 2>&1 | tail -1)"
done | tee "$OUT/r04i_mx_step_times.txt"
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
  local db; db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r04_${name}_kernel_stats.csv" > /dev/null
  rm -rf "$OUT/prof_$name"
  echo "== $name: $(grep ms/step "$OUT/prof_$name.log")"; grep -E "gemv|attn|norm" "$OUT/r04_${name}_kernel_stats.csv" | head -10 | cut -c1-160
}
prof batch64_fp8_mx python "$REPO/tools/bench_batch.py" --batch 64 --fork --steps 16 --model detikzify-cl-7b --weight-format fp8
cd "$REPO"
timeout 1500 python -m pytest tests/test_gpu_parity_batched.py -q -p no:cacheprovider -s --tb=short -k "(headline and cl-7b) or v2_8b or peaked" 2>&1 | grep -E "^batched|^peaked|passed|failed|Error|assert|^E " | cut -c1-2000 | tee "$OUT/r04i_full_depth.txt"
