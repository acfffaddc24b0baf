#!/usr/bin/env bash
# round 4, lease A: the multi-vector step (tests at toy size, step times against the MFMA family and the single-sequence graph,
# block-shape sweep), the existing small-slot tests pinned to the MFMA family, kernel traces of the 2- and 16-slot steps, and the
# fp8 64-slot step with counters (what bounds it).
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
echo "== multi-vector tests"
timeout 900 python -m pytest tests/test_gpu_parity_mv.py -q -x -p no:cacheprovider -s 2>&1 | grep -v "^$" | tail -25
echo "== small-slot tests of the MFMA family"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider --durations=6 \
  -k "batched_decode_tracks or batch_engine_threads or resume_slot or kv_fork or fp8_weights or engine_prefix or shared_prefix_reads or context_limits or simulate_parallel or several_images" 2>&1 | tail -14
echo "== tune_mv ds-7b"
timeout 600 python tools/tune_mv.py --model detikzify-ds-7b > "$OUT/r04_tune_mv_ds7b.txt" 2>&1; cat "$OUT/r04_tune_mv_ds7b.txt" | tail -30
echo "== 16 / 8 slot MFMA steps"
for b in 8 16; do timeout 300 python tools/bench_batch.py --batch $b --slots 17 --fork --steps 48 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
prof() {   # name, then the command's arguments
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
  local db; db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r04_${name}_kernel_stats.csv" > /dev/null
  rm -rf "$OUT/prof_$name"
  echo "== $name: $(grep ms/step "$OUT/prof_$name.log")"; head -14 "$OUT/r04_${name}_kernel_stats.csv" | cut -c1-150
}
prof mv2 python "$REPO/tools/bench_batch.py" --batch 2 --slots 3 --fork --steps 24
prof mv4 python "$REPO/tools/bench_batch.py" --batch 4 --slots 5 --fork --steps 24
prof batch16 python "$REPO/tools/bench_batch.py" --batch 16 --slots 17 --fork --steps 24
prof batch64_fp8 python "$REPO/tools/bench_batch.py" --batch 64 --fork --steps 16 --model detikzify-cl-7b --weight-format fp8
pmc() {   # name, counters..., -- command
  local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout 400 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" -d "$OUT/pmc_$name" -o pmc -- "$@" > "$OUT/pmc_$name.log" 2>&1
  echo "rocprof pmc $name exit $?"
  local db; db=$(ls "$OUT"/pmc_$name/*/*.db "$OUT"/pmc_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r04_${name}.csv" --pmc > /dev/null
  rm -rf "$OUT/pmc_$name"
  grep -E "gemv|attn|norm" "$OUT/r04_${name}.csv" | cut -c1-200 | head -40
}
F8CMD=(python "$REPO/tools/bench_batch.py" --batch 64 --fork --steps 6 --model detikzify-cl-7b --weight-format fp8)
pmc batch64_fp8_pmc_fetch FETCH_SIZE -- "${F8CMD[@]}"
pmc batch64_fp8_pmc_sq SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -- "${F8CMD[@]}"
BFCMD=(python "$REPO/tools/bench_batch.py" --batch 64 --fork --steps 6)
pmc batch64_bf16_pmc_sq SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -- "${BFCMD[@]}"
