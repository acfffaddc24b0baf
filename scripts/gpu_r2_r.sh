#!/usr/bin/env bash
# round 2, call R: where the parallel MCTS phase of bench.py spends its wall time (engine idle / occupancy diagnostics)
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
DTK_TRACE_MCTS=$OUT/r2r_trace.json timeout 900 python bench.py --steps 1 --warmup 0 --skip-batched --no-cpu-baseline --probe-tokens 2 > "$OUT/r2r_bench.log" 2> "$OUT/r2r_bench.err"; echo "bench exit $?"
python - <<'PY'
import json
for ln in open("gpurun_out/r2r_bench.log"):
    if ln.startswith("{"):
        d = json.loads(ln); m = d["mcts"]
        print(json.dumps(m["parallel"])); print(json.dumps(m["sequential"]))
PY
python tools/mcts_timeline.py gpurun_out/r2r_trace.json
