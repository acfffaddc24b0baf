#!/usr/bin/env bash
# round 3, lease L: the bench lines again on the final kernels (default line with CPU baselines and configs 4 / 5; the other families)
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 1500 python bench.py > "$OUT/r03_bench_ds7b.json" 2> "$OUT/r3l_bench.err"; echo "bench exit $?"
for cfg in "detikzify-ds-1.3b bf16" "detikzify-cl-7b fp8" "detikzify-v2-8b bf16"; do
  set -- $cfg
  timeout 900 python bench.py --model $1 --weight-format $2 --no-cpu-baseline --no-config5 --steps 2 > "$OUT/r03_bench_${1#detikzify-}_$2.json" 2>/dev/null; echo "$1 $2 exit $?"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r03_bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    b = d.get("batched_rollouts") or {}; m = d.get("mcts") or {}
    c4 = ((m.get("config4") or {}).get("fixed_length") or {}).get("rollouts_per_sec"); c5 = ((m.get("config5") or {}).get("fixed_length") or {}).get("rollouts_per_sec")
    print(f.split("/")[-1], "tok/s", round(d["value"], 1), "decode", round(d["decode_tokens_per_sec_per_gpu"], 1), "frac", round(d["decode_step"]["frac_of_hbm_peak"], 3),
          "| batched", round(b.get("rollouts_per_sec", 0), 2), round(b.get("frac_of_hbm_peak", 0), 3), "| mcts seq", round((m.get("sequential") or {}).get("rollouts_per_sec", 0) or 0, 3),
          "par", round((m.get("parallel") or {}).get("rollouts_per_sec", 0) or 0, 2), "c4", c4, "c5", c5, "| roofline", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"),
          "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"), (d.get("cpu_baseline") or {}).get("parity_tokens_identical"))
PY
