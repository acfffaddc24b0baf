#!/usr/bin/env bash
# round 5, lease C: the batched attention after the dependent-load chains of k_attn_tail_b / k_attn_prefix_g were cut (every scalar
# and q requested together; all prefix-state splits in flight at once): step times with and without the grouped prefix kernel, kernel
# trace of the 64-slot step, then the attention tests and the restructured MXFP8-vs-bf16-activations test.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
SB=$REPO/tools/probe/step_bench
{
echo "== 64 slots, cl-7b fp8 (bf16 activations = the default), 1 image"
STEP_BENCH_SLOTS=64 timeout 300 $SB "prefix_mfma=0" "" "tail_threads=128" "prefix_mfma=0,tail_threads=128" "act_fp8=1,prefix_mfma=0" "act_fp8=1"
echo "== 64 slots, cl-7b fp8, 8 images x 8 forks (BASELINE config 5 on one GPU)"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 timeout 300 $SB "prefix_mfma=0" "" "act_fp8=1,prefix_mfma=0" "act_fp8=1"
echo "== 64 slots, cl-7b fp8, 1 image, 260 private keys per slot"
STEP_BENCH_SLOTS=64 STEP_BENCH_WARM=260 timeout 300 $SB "prefix_mfma=0" "" "tail_threads=128"
echo "== 64 / 32 / 16 slots, ds-7b bf16, 1 image"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 timeout 300 $SB "prefix_mfma=0" ""
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=32 timeout 200 $SB "prefix_mfma=0" ""
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=16 timeout 200 $SB "prefix_mfma=0" ""
} 2>&1 | tee "$OUT/r05c_step_bench.txt"
cd /tmp && export TMPDIR=/tmp
prof() {   # name, env..., -- command
  local name=$1; shift
  timeout 300 env "$@" rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- $SB "$PROF_OPTS" > "$OUT/prof_$name.log" 2>&1
  local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r05c_$name.csv" > /dev/null
  rm -rf "$OUT/prof_$name"; echo "-- $name [$PROF_OPTS]"; head -9 "$OUT/r05c_$name.csv" | cut -c1-170
}
PROF_OPTS="" prof batch64_fp8_prefix_kernel_stats STEP_BENCH_SLOTS=64
PROF_OPTS="prefix_mfma=0" prof batch64_fp8_noprefix_kernel_stats STEP_BENCH_SLOTS=64
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_parity_attn.py "tests/test_gpu_parity.py::test_shared_prefix_on_matrix_cores_tracks_the_per_slot_path" tests/test_gpu_parity_batched.py::test_mxfp8_activations_against_bf16_activations -q -p no:cacheprovider -rA --tb=short > "$OUT/r05c_tests.log" 2>&1
echo "pytest exit $?"; grep -E "passed|failed" "$OUT/r05c_tests.log" | tail -1; grep -E "^FAILED|^ERROR|^E  |^MXFP8|^grouped" "$OUT/r05c_tests.log" | cut -c1-1200 | head -30
