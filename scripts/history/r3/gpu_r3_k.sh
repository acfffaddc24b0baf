#!/usr/bin/env bash
# round 3, lease K: after the launch-bound change of k_gemv_bl (accumulators in VGPRs) — the batched-path GPU tests, the three 64-slot
# step profiles again (kernel names changed), smoke()
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_batched.py -q -x -p no:cacheprovider -k "batch or slot or x_once or fp8 or peaked or engine or parallel or gqa" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1 ctrs=$2; shift 2
  if [ -z "$ctrs" ]; then timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
  else timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/prof_$name" -o pmc -- "$@" > "$OUT/prof_$name.log" 2>&1; fi
  local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r03_$name.csv" $([ -n "$ctrs" ] && echo --pmc) > /dev/null
  rm -rf "$OUT/prof_$name"; grep ms/step "$OUT/prof_$name.log"; head -7 "$OUT/r03_$name.csv" | cut -c1-150
}
prof batch64_kernel_stats "" python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork
prof batch64_pmc_fetch "FETCH_SIZE" python "$REPO/tools/bench_batch.py" --batch 64 --steps 8 --fork
prof batch64_ctx500_kernel_stats "" python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --ctx 500 --private
cd "$REPO"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
