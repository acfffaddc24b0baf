#!/usr/bin/env bash
# round 4, lease H: the fp8 matrix-core step (csrc/kernels_batch_mx.hip) for the first time — op-level tests (quantiser + fragment
# order bit-exact, the two GEMV kernels and the SwiGLU epilogue against float64), the two-layer step against the quantising oracle,
# step times with act_fp8 = 1 / 0 at 64 / 32 / 16 slots, a kernel trace of the 64-slot step, then the full-depth cl-7b fp8 test.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity_mx.py -q -p no:cacheprovider -s --tb=short 2>&1 | grep -vE "amdgpu.ids|^$" | cut -c1-1200 | tail -90 | tee "$OUT/r04h_mx_tests.txt"
for b in 64 32 16; do
  for o in 1 0; do
    echo "-- cl-7b fp8, $b slots, act_fp8=$o: $(DTK_OPTIONS=act_fp8=$o timeout 300 python tools/bench_batch.py --batch $b --fork --steps 32 --model detikzify-cl-7b --weight-format fp8 2>&1 | tail -1)"
  done
done | tee "$OUT/r04h_mx_step_times.txt"
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
timeout 1500 python -m pytest tests/test_gpu_parity_batched.py -q -p no:cacheprovider -s --tb=short -k "headline and cl-7b" 2>&1 | grep -E "^batched|passed|failed|Error|assert|^E " | cut -c1-1800 | tee "$OUT/r04h_mx_full_depth.txt"
