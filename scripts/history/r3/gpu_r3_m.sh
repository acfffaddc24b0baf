#!/usr/bin/env bash
# round 3, lease M: f3 with more trees than decode slots — bench.py --reward-latency 1 5 (1 tree, 64 trees, 128 trees over 64 slots)
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 1200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --skip-batched --no-config4 --no-config5 --mcts-seq-expansions 0 --probe-tokens 4 \
  --reward-latency 1 5 > "$OUT/r03_bench_reward_latency.json" 2> "$OUT/r3m.err"; echo "exit $?"
python - <<PY
import json
d = json.loads([l for l in open("$OUT/r03_bench_reward_latency.json") if l.startswith("{")][-1])
for S, e in d["mcts"]["reward_latency"].items():
    for k, v in e.items():
        print(S, k, v["rollouts"], round(v["rollouts_per_sec"], 2), round(v["seconds"], 1), v.get("engine_wait_s"))
print("parallel", d["mcts"]["parallel"]["rollouts_per_sec"])
PY
