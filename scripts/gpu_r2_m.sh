#!/usr/bin/env bash
# round 2, call M: k_gemv_bk (K split across CUs for o_proj / down) — bit-identity, per-kernel profile, batched bench phase
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "x_once_per_cu" > "$OUT/r2m_pytest.log" 2>&1
echo "pytest exit $?"; tail -15 "$OUT/r2m_pytest.log" | cut -c1-300
cd /tmp && export TMPDIR=/tmp
DTK_OPTIONS="gemv_bx=1,gemv_bk=1" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_bk" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork > "$OUT/prof_bk.log" 2>&1; echo "rocprof exit $?"
python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_bk -name trace_results.db | head -1)" "$OUT/r02_batch64_bk_kernel_stats.csv" > /dev/null 2>&1
rm -rf "$OUT/prof_bk"; tail -2 "$OUT/prof_bk.log" | cut -c1-200; head -7 "$OUT/r02_batch64_bk_kernel_stats.csv" | cut -c1-150
cd "$REPO"
for bk in 0 1; do
  DTK_OPTIONS="gemv_bk=$bk" timeout 600 python bench.py --steps 1 --warmup 1 --mcts-trees 0 --no-cpu-baseline > "$OUT/r2m_bench_bk$bk.log" 2> "$OUT/r2m_bench_bk$bk.err"; echo "bench bk=$bk exit $?"
  python - "$OUT/r2m_bench_bk$bk.log" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln); b = d["batched_rollouts"]
        print("value", round(d["value"], 1), "batched rollouts/s", round(b["rollouts_per_sec"], 2), "frac", round(b["frac_of_hbm_peak"], 3), "ms/batch", round(b["ms_per_batch"]), "steps", b["decode_steps"])
PY
done
