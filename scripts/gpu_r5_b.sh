#!/usr/bin/env bash
# round 5, lease B: (1) the attention-launch weight prefetch of the single-sequence step (option attn_prefetch) and (2) the grouped
# shared-prefix attention of the batched step (k_attn_prefix_g, default on) timed through the Python-free harness, with kernel traces;
# (3) the new attention tests + the two tests lease A failed (long context: one flip at 2.30 ulps; MXFP8 figures: a draw-agreement
# assertion that does not hold on the uniform head).
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
SB=$REPO/tools/probe/step_bench
{
echo "== single sequence, ds-7b bf16 (BASELINE headline step): attention-launch prefetch of o_proj's weights"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=1 STEP_BENCH_STEPS=256 STEP_BENCH_WARM=16 timeout 300 $SB "" "attn_prefetch=1" "attn_prefetch=2" "attn_prefetch=1,attn_prefetch_blocks=2" "attn_prefetch=1,attn_prefetch_blocks=8" ""
echo "== single sequence, cl-7b fp8 and ds-1.3b"
STEP_BENCH_MODEL=cl-7b-fp8 STEP_BENCH_SLOTS=1 STEP_BENCH_STEPS=256 STEP_BENCH_WARM=16 timeout 200 $SB "" "attn_prefetch=1" "attn_prefetch=1,attn_prefetch_blocks=8"
STEP_BENCH_MODEL=ds-1.3b STEP_BENCH_SLOTS=1 STEP_BENCH_STEPS=256 STEP_BENCH_WARM=16 timeout 200 $SB "" "attn_prefetch=1" "attn_prefetch=1,attn_prefetch_blocks=8"
echo "== 64 slots, cl-7b fp8 (bf16 activations = the default), 1 image: grouped prefix attention"
STEP_BENCH_SLOTS=64 timeout 300 $SB "prefix_mfma=0" "" "pfx_splits=2" "act_fp8=1,prefix_mfma=0" "act_fp8=1"
echo "== 64 slots, cl-7b fp8, 8 images x 8 forks (BASELINE config 5 on one GPU)"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 timeout 300 $SB "prefix_mfma=0" "" "act_fp8=1,prefix_mfma=0" "act_fp8=1"
echo "== 64 slots, cl-7b fp8, 1 image, 260 private keys per slot"
STEP_BENCH_SLOTS=64 STEP_BENCH_WARM=260 timeout 300 $SB "prefix_mfma=0" ""
echo "== 64 slots, ds-7b bf16, 1 image"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 timeout 300 $SB "prefix_mfma=0" ""
echo "== 16 and 32 slots, ds-7b bf16, 1 image"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=16 timeout 200 $SB "prefix_mfma=0" ""
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=32 timeout 200 $SB "prefix_mfma=0" ""
} 2>&1 | tee "$OUT/r05b_step_bench.txt"
cd /tmp && export TMPDIR=/tmp
prof() {   # name, env..., -- command
  local name=$1; shift
  timeout 300 env "$@" rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- $SB "$PROF_OPTS" > "$OUT/prof_$name.log" 2>&1
  local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r05b_$name.csv" > /dev/null
  rm -rf "$OUT/prof_$name"; echo "-- $name [$PROF_OPTS]"; head -9 "$OUT/r05b_$name.csv" | cut -c1-170
}
PROF_OPTS="" prof single_ds7b_kernel_stats STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=1 STEP_BENCH_STEPS=128
PROF_OPTS="attn_prefetch=1" prof single_ds7b_prefetch_kernel_stats STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=1 STEP_BENCH_STEPS=128
PROF_OPTS="" prof batch64_fp8_prefix_kernel_stats STEP_BENCH_SLOTS=64
PROF_OPTS="prefix_mfma=0" prof batch64_fp8_noprefix_kernel_stats STEP_BENCH_SLOTS=64
PROF_OPTS="" prof batch64_fp8_8img_prefix_kernel_stats STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8
cd "$REPO"
timeout 900 python -m pytest tests/test_gpu_parity_attn.py "tests/test_gpu_parity.py::test_shared_prefix_on_matrix_cores_tracks_the_per_slot_path" tests/test_gpu_parity_batched.py::test_long_context_steps_match_cpu_oracle tests/test_gpu_parity_batched.py::test_mxfp8_activations_against_bf16_activations -q -p no:cacheprovider -rA --tb=short > "$OUT/r05b_tests.log" 2>&1
echo "pytest exit $?"; grep -E "passed|failed" "$OUT/r05b_tests.log" | tail -1; grep -E "^FAILED|^ERROR|^E  |^MXFP8|^grouped|long-context" "$OUT/r05b_tests.log" | cut -c1-900 | head -30
