#!/usr/bin/env bash
# round 2, call O: batched phase of bench.py for tail block size x shared-prefix MFMA kernel
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
for opts in "tail_threads=256,prefix_mfma=0" "tail_threads=128,prefix_mfma=0" "tail_threads=128,prefix_mfma=1,pfx_splits=2" "tail_threads=128,prefix_mfma=1,pfx_splits=1" "tail_threads=64,prefix_mfma=1,pfx_splits=2"; do
  DTK_OPTIONS="$opts" timeout 600 python bench.py --steps 1 --warmup 1 --mcts-trees 0 --no-cpu-baseline > "$OUT/r2o_bench.log" 2> "$OUT/r2o_bench.err"
  python - "$OUT/r2o_bench.log" "$opts" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln); b = d["batched_rollouts"]
        print(sys.argv[2], "| batched rollouts/s", round(b["rollouts_per_sec"], 2), "frac", round(b["frac_of_hbm_peak"], 3), "ms/batch", round(b["ms_per_batch"]))
PY
done
