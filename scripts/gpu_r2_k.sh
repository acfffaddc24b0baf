#!/usr/bin/env bash
# round 2, call K: the batched phase of bench.py with and without k_gemv_bx, per-kernel times of the 64-slot step
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
for bx in 0 1; do
  DTK_OPTIONS="gemv_bx=$bx" timeout 600 python bench.py --steps 1 --warmup 1 --mcts-trees 0 --no-cpu-baseline > "$OUT/r2k_bench_bx$bx.log" 2> "$OUT/r2k_bench_bx$bx.err"; echo "bench bx=$bx exit $?"
  python - "$OUT/r2k_bench_bx$bx.log" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln); b = d["batched_rollouts"]
        print("value", round(d["value"], 1), "batched rollouts/s", round(b["rollouts_per_sec"], 2), "frac", round(b["frac_of_hbm_peak"], 3), "ms/batch", round(b["ms_per_batch"]), "steps", b["decode_steps"])
PY
done
cd /tmp && export TMPDIR=/tmp
DTK_OPTIONS="gemv_bx=1" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_bx" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork > "$OUT/prof_bx.log" 2>&1; echo "rocprof exit $?"
python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_bx -name trace_results.db | head -1)" "$OUT/r02_batch64_bx_kernel_stats.csv" > /dev/null 2>&1
rm -rf "$OUT/prof_bx"; tail -2 "$OUT/prof_bx.log"; head -12 "$OUT/r02_batch64_bx_kernel_stats.csv" | cut -c1-150
