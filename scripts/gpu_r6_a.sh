#!/usr/bin/env bash
# round 6, lease A: the native run loop (dtk_engine_*) on the device for the first time — the GPU tests that touch an engine, then the
# default bench line with the native loop and, for the A/B, with the Python-driven engine (DTK_ENGINE=python).
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
timeout 900 python -m pytest tests -m gpu -x -q -k "engine or kv_fork or resume or parallel or simulate or smoke or abi" 2>&1 | tail -15 | tee "$OUT/r06a_pytest_engine.txt"
timeout 900 python bench.py --no-cpu-baseline > "$OUT/r06a_bench_native.json" 2> "$OUT/r06a_bench_native.err"; echo "bench native rc $?"
DTK_ENGINE=python timeout 900 python bench.py --no-cpu-baseline > "$OUT/r06a_bench_python_engine.json" 2> "$OUT/r06a_bench_python_engine.err"; echo "bench python rc $?"
python - <<'PY'
import json
for name in ("native", "python_engine"):
    try:
        p = json.loads(open(f"gpurun_out/r06a_bench_{name}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "unreadable:", e); continue
    m = p.get("mcts", {})
    print(name, "tok/s", round(p["value"], 1), "batched", round(p.get("batched_rollouts", {}).get("rollouts_per_sec", 0), 2))
    for k in ("parallel", "parallel_oversubscribed"):
        r = m.get(k) or {}
        print("  ", k, r.get("rollouts_per_sec"), r.get("seconds"), r.get("engine"))
    for cfg in ("config4", "config5"):
        for k in ("fixed_length", "ragged"):
            r = (m.get(cfg) or {}).get(k) or {}
            print("  ", cfg, k, r.get("rollouts_per_sec"), r.get("seconds"), r.get("engine"))
PY
