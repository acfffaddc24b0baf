#!/usr/bin/env bash
# Full GPU parity suite, then the MCTS phase of bench.py (64 trees x 3 expansions, stub reward) — one box, ~2 minutes.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 110 python -m pytest tests -m gpu -q --tb=short -s -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?"
{ grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -1; grep -E "^FAILED|^ERROR" "$OUT/pytest_gpu.log"
  sed -E 's/^[.sFE]+//' "$OUT/pytest_gpu.log" | grep -vE "^$|passed|failed|[Ww]arning|^  |^=|^-|amdgpu.ids" | head -300; } > "$OUT/pytest_gpu_summary.txt"
head -2 "$OUT/pytest_gpu_summary.txt"; grep -E "^E  " "$OUT/pytest_gpu.log" | head -5
timeout 80 python bench.py --steps 1 --warmup 0 --new-tokens 256 --no-cpu-baseline --batch 64 --skip-batched --mcts-trees 64 --mcts-expansions 3 --probe-tokens 2 \
  > "$OUT/bench_mcts.log" 2> "$OUT/bench_mcts.err"
echo "bench exit $?"; python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_mcts.log").read().strip().splitlines()[-1]); print(json.dumps(d.get("mcts_stub_reward")))
except Exception as e:
    print("no bench line", e)
PY
