#!/usr/bin/env bash
# round 3, lease N: the default bench line with mcts.parallel_oversubscribed (96 trees taking turns in 64 decode slots)
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 1500 python bench.py > "$OUT/r03_bench_ds7b.json" 2> "$OUT/r3n_bench.err"; echo "bench exit $?"
python - <<PY
import json
d = json.loads([l for l in open("$OUT/r03_bench_ds7b.json") if l.startswith("{")][-1])
m = d["mcts"]
print("tok/s", round(d["value"], 1), "decode", round(d["decode_tokens_per_sec_per_gpu"], 1), "batched", round(d["batched_rollouts"]["rollouts_per_sec"], 2),
      "seq", round(m["sequential"]["rollouts_per_sec"], 3), "par", round(m["parallel"]["rollouts_per_sec"], 2),
      "over", round(m["parallel_oversubscribed"]["rollouts_per_sec"], 2), m["parallel_oversubscribed"]["rollouts"], m["parallel_oversubscribed"].get("frac_of_roofline"),
      "c4", m["config4"]["fixed_length"]["rollouts_per_sec"], m["config4"]["ragged"]["rollouts_per_sec"], "c5", m["config5"]["fixed_length"]["rollouts_per_sec"], m["config5"]["ragged"]["rollouts_per_sec"])
print(json.dumps(m["parallel_oversubscribed"].get("engine")))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["parity_tokens_identical"], "roofline", d["roofline"]["frac"], d["roofline"]["traffic"])
PY
