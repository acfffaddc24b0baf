#!/usr/bin/env bash
# round 2, last call: the whole GPU suite + smoke + the default bench line on the final tree
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > "$OUT/r2fin_pytest.log" 2>&1; echo "pytest exit $?"; tail -2 "$OUT/r2fin_pytest.log"
grep -E "ViT per-block|decoder depth|resume in place|GQA fused|dev-vs-bf16-oracle|passed|failed" "$OUT/r2fin_pytest.log" | cut -c1-600 > "$OUT/r02_pytest_gpu_summary.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> "$OUT/r02_pytest_gpu_summary.txt"
timeout 1500 python bench.py > "$OUT/r02_bench_ds7b.json" 2> "$OUT/r2fin_bench.err"; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r02_bench_ds7b.json") if l.startswith("{")][-1])
b = d["batched_rollouts"]; m = d["mcts"]; r = d["roofline"]; c = d["cpu_baseline"]
print("value", round(d["value"], 1), "decode", round(d["decode_tokens_per_sec_per_gpu"], 1), round(d["decode_step"]["frac_of_hbm_peak"], 3), "| batched", round(b["rollouts_per_sec"], 2), round(b["frac_of_hbm_peak"], 3),
      "| mcts", round(m["sequential"]["rollouts_per_sec"], 3), round(m["parallel"]["rollouts_per_sec"], 2), "| roofline", round(r["frac"], 3), r["avg_launch_us"], r["traffic"], "| cpu", c["value"], c["cores"], c["parity_tokens_identical"])
PY
