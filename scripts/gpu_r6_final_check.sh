#!/usr/bin/env bash
# round 6 — what the driver runs at round end, on HEAD: the GPU suite, smoke(), the flag-less bench
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06head}
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=5 ) 2>&1 | tail -15 | cut -c1-200 | tee "$OUT/${R}_pytest_gpu.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee "$OUT/${R}_smoke.txt"
( time timeout 1500 python bench.py > "$OUT/${R}_bench_default.json" 2> "$OUT/${R}_bench_default.err" ) 2>&1 | grep real
python - "$OUT/${R}_bench_default.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
m = d.get("mcts") or {}
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "scaling", "vs_baseline")})
print("roofline", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], "prefill", d["prefill_ms"], "vit", d["vit_ms"],
      "| par", (m.get("parallel") or {}).get("rollouts_per_sec"), "c4", ((m.get("config4") or {}).get("fixed_length") or {}).get("rollouts_per_sec"),
      "c5", ((m.get("config5") or {}).get("fixed_length") or {}).get("rollouts_per_sec"), ((m.get("config5") or {}).get("ragged") or {}).get("rollouts_per_sec"))
PY
