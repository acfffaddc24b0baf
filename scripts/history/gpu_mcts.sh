#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 64 --mcts-trees 64 --mcts-expansions 3 --probe-tokens 8 > "$OUT/bench_mcts.log" 2> "$OUT/bench_mcts.err"; tail -2 "$OUT/bench_mcts.err" | cut -c1-300
python - <<PY
import json
d=json.loads(open("$OUT/bench_mcts.log").read().strip().splitlines()[-1]); print(json.dumps(d.get("mcts_stub_reward"))); print("batched", round(d["batched_rollouts"]["rollouts_per_sec"],2))
PY
