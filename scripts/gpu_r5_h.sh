#!/usr/bin/env bash
# round 5, lease H — the closing run on the FINAL source (after lease G made the batched attention's shape a property of the context's
# size, the round-1 attention path was pruned and bench.py's config 4 got a context of its own size): the whole GPU suite as the
# driver runs it, smoke(), the default bench line.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
( time timeout 1500 python -m pytest tests -m gpu -q --tb=short -rA -p no:cacheprovider --durations=25 ) > "$OUT/r05_pytest_gpu.log" 2>&1
echo "pytest exit $?"
{ grep -E "passed|failed" "$OUT/r05_pytest_gpu.log" | tail -1; grep -E "^real" "$OUT/r05_pytest_gpu.log"; grep -E "^FAILED|^ERROR" "$OUT/r05_pytest_gpu.log"
  echo "== slowest"; grep -E "^[0-9.]+s (call|setup)" "$OUT/r05_pytest_gpu.log" | head -25
  echo "== figures printed by the tests"
  grep -vE "^$|^PASSED|^SKIPPED|^=|^-|^_|amdgpu.ids|[Ww]arning|^  |^[.sFE]+ +\[|^[0-9.]+s (call|setup)|Captured stdout" "$OUT/r05_pytest_gpu.log" | cut -c1-2500 | head -400; } > "$OUT/r05_pytest_gpu_summary.txt"
head -3 "$OUT/r05_pytest_gpu_summary.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee "$OUT/r05_smoke.txt"
timeout 1500 python bench.py --steps 20 --warmup 5 > "$OUT/r05_bench_ds7b.json" 2> "$OUT/r05_bench.err"; echo "bench exit $?"
python - <<PY
import json
d = json.loads([l for l in open("$OUT/r05_bench_ds7b.json") if l.startswith("{")][-1])
m = d["mcts"]; c4 = m["config4"]; c5 = m["config5"]
print("tok/s", round(d["value"], 1), "| batched", round(d["batched_rollouts"]["rollouts_per_sec"], 2), "| mcts seq", round(m["sequential"]["rollouts_per_sec"], 3), "par", round(m["parallel"]["rollouts_per_sec"], 2),
      "over", round(m["parallel_oversubscribed"]["rollouts_per_sec"], 2), "| c4", round(c4["fixed_length"]["rollouts_per_sec"], 2), round(c4["ragged"]["rollouts_per_sec"], 2), "ctx", c4.get("context_slots"),
      {k: round(v["rollouts_per_sec_one_rank"], 2) for k, v in c4["rank_shape"].items() if isinstance(v, dict)},
      "| c5", round(c5["fixed_length"]["rollouts_per_sec"], 2), round(c5["ragged"]["rollouts_per_sec"], 2), round(c5["fixed_length_fp8_matrix_cores_opt_in"]["rollouts_per_sec"], 2),
      "| roofline", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], "| predicted", d.get("mcts_config4_predicted_rollouts_per_sec"), d.get("mcts_config5_predicted_rollouts_per_sec"))
PY
