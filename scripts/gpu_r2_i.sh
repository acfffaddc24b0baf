#!/usr/bin/env bash
# round 2, call I: per-block / per-depth error-growth tests, batched gate/up probes (occupancy, pipelining, x traffic), roofline leg
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -s -k "error_grows" > "$OUT/r2i_pytest.log" 2>&1
echo "pytest exit $?"; grep -E "ViT per-block|decoder depth|passed|failed|Error|assert" "$OUT/r2i_pytest.log" | cut -c1-400 | tail -20
timeout 600 python tools/probe_batch.py --no-lds > "$OUT/r2i_probe_batch.log" 2>&1; echo "probe exit $?"; grep slots "$OUT/r2i_probe_batch.log"
timeout 600 python bench.py --steps 2 --warmup 1 --batch 0 --mcts-trees 0 --no-cpu-baseline > "$OUT/r2i_bench.log" 2> "$OUT/r2i_bench.err"; echo "bench exit $?"
python - <<'PY'
import json
for ln in open("gpurun_out/r2i_bench.log"):
    if ln.startswith("{"):
        d = json.loads(ln); print(json.dumps(d["roofline"])); print(d["value"], d["decode_tokens_per_sec_per_gpu"])
PY
