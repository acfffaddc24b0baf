#!/usr/bin/env bash
# round 6, lease Y — the four tests lease X failed (three near-tie rules met by another fp32 summation order of the prefix), then the default bench line
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06y}
timeout 1200 python -m pytest tests -m gpu -q -k "prefill_kernel_switches or peaked_logits or greedy_margins" -rA 2>&1 | grep -v "^PASSED\|^SKIPPED" | tail -40 | cut -c1-330 > "$OUT/${R}_pytest.txt"; grep -n "^E \|passed\|failed\|^FAILED\|peaked weight set\|margin sign" "$OUT/${R}_pytest.txt" | cut -c1-330 | head -20
timeout 1500 python bench.py > "$OUT/${R}_bench_ds7b.json" 2> "$OUT/${R}_bench.err"; tail -c 1500 "$OUT/${R}_bench_ds7b.json" | head -c 600; echo
python - "$OUT/${R}_bench_ds7b.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "prefill_ms", d.get("prefill_ms"), "vit_ms", d.get("vit_ms"))
m = d.get("mcts", {})
for k in ("sequential", "parallel", "parallel_oversubscribed"):
    print(k, (m.get(k) or {}).get("rollouts_per_sec"))
for k in ("config4", "config5"):
    v = m.get(k) or {}
    print(k, {kk: (vv.get("rollouts_per_sec") if isinstance(vv, dict) else None) for kk, vv in v.items() if kk in ("fixed_length", "ragged", "fixed_length_fp8_matrix_cores_opt_in")})
PY
