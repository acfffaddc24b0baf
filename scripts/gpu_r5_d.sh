#!/usr/bin/env bash
# round 5, lease D: k_attn_tail_b no longer loads rows beyond the context — step times with / without the grouped prefix kernel at
# several private-context lengths (4, 36, 132, 260: the old clamp wasted most at p mod 64 = 4), block sizes 128 / 256, kernel trace.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
SB=$REPO/tools/probe/step_bench
{
for warm in 4 36 132 260; do
  echo "== 64 slots, cl-7b fp8 (bf16 activations), 1 image, $warm private keys"
  STEP_BENCH_SLOTS=64 STEP_BENCH_WARM=$warm timeout 300 $SB "prefix_mfma=0" "" "tail_threads=128" "prefix_mfma=0,tail_threads=128"
done
echo "== 64 slots, cl-7b fp8, 8 images x 8 forks (BASELINE config 5 on one GPU), 4 and 132 private keys"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 timeout 300 $SB "prefix_mfma=0" "" "tail_threads=128" "act_fp8=1,prefix_mfma=0" "act_fp8=1" "act_fp8=1,tail_threads=128"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_WARM=132 timeout 300 $SB "prefix_mfma=0" "" "tail_threads=128"
echo "== 64 / 32 / 16 slots, ds-7b bf16, 1 image"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 timeout 300 $SB "prefix_mfma=0" "" "tail_threads=128"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=32 timeout 200 $SB "prefix_mfma=0" "" "tail_threads=128"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=16 timeout 200 $SB "prefix_mfma=0" "" "tail_threads=128"
} 2>&1 | tee "$OUT/r05d_step_bench.txt"
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift
  timeout 300 env "$@" rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- $SB "$PROF_OPTS" > "$OUT/prof_$name.log" 2>&1
  local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r05d_$name.csv" > /dev/null
  rm -rf "$OUT/prof_$name"; echo "-- $name [$PROF_OPTS]"; head -8 "$OUT/r05d_$name.csv" | cut -c1-150
}
PROF_OPTS="" prof batch64_fp8_prefix_kernel_stats STEP_BENCH_SLOTS=64
PROF_OPTS="prefix_mfma=0" prof batch64_fp8_noprefix_kernel_stats STEP_BENCH_SLOTS=64
PROF_OPTS="tail_threads=128" prof batch64_fp8_prefix_t128_kernel_stats STEP_BENCH_SLOTS=64
