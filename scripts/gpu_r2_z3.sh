#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
for tt in 1 2; do
  DTK_OPTIONS="gqa_fused=$tt" timeout 900 python bench.py --model detikzify-v2-8b --steps 1 --warmup 1 --mcts-trees 0 --no-cpu-baseline > "$OUT/r2z_bench.log" 2> "$OUT/r2z_bench.err"
  python - "$OUT/r2z_bench.log" "gqa_fused=$tt" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln); b = d["batched_rollouts"]
        print(sys.argv[2], "| batched rollouts/s", round(b["rollouts_per_sec"], 2), "frac", round(b["frac_of_hbm_peak"], 3), "ms/batch", round(b["ms_per_batch"]))
PY
done
