#!/usr/bin/env bash
# round 2, call Z2: v2-8b with the GQA-fused attention: all v2 tests, full-size parity, per-kernel profile of the batched phase, default bench line
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -s -k "gqa_heads or v2" > "$OUT/r2z2_pytest.log" 2>&1
echo "pytest exit $?"; grep -E "GQA fused|v2-8b|passed|failed" "$OUT/r2z2_pytest.log" | cut -c1-300 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_z" -o trace -- python "$REPO/bench.py" --model detikzify-v2-8b --steps 1 --warmup 0 --mcts-trees 0 --no-cpu-baseline --probe-tokens 2 > "$OUT/prof_z.log" 2>&1; echo "rocprof exit $?"
python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_z -name trace_results.db | head -1)" "$OUT/r02_bench_batched_phase_v2_8b_kernel_stats.csv" > /dev/null 2>&1
rm -rf "$OUT/prof_z"; head -12 "$OUT/r02_bench_batched_phase_v2_8b_kernel_stats.csv" | cut -c1-150
cd "$REPO"
timeout 900 python bench.py --model detikzify-v2-8b --weight-format bf16 --no-cpu-baseline --steps 2 > "$OUT/r02_bench_v2-8b_bf16.json" 2>/dev/null; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r02_bench_v2-8b_bf16.json") if l.startswith("{")][-1])
b = d["batched_rollouts"]; m = d["mcts"]
print("v2-8b value", round(d["value"], 1), "decode", round(d["decode_tokens_per_sec_per_gpu"], 1), "| batched", round(b["rollouts_per_sec"], 2), round(b["frac_of_hbm_peak"], 3), "| mcts", round(m["sequential"]["rollouts_per_sec"], 3), round(m["parallel"]["rollouts_per_sec"], 2))
PY
