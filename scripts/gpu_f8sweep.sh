#!/usr/bin/env bash
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
for v in 0 1 2 3 4 5; do
  DTK_F8_VARIANT=$v timeout 300 python bench.py --model detikzify-cl-7b --weight-format fp8 --steps 1 --warmup 1 --no-cpu-baseline --batch 0 --probe-tokens 16 > "$OUT/bench_f8_v$v.log" 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$OUT/bench_f8_v$v.log").read().strip().splitlines()[-1]); print("f8 variant $v: decode tok/s", round(d["decode_tokens_per_sec_per_gpu"],1), "gate/up us", round(d["roofline"].get("avg_launch_us",0),2))
PY
done
cd /tmp && export TMPDIR=/tmp
DTK_F8_VARIANT=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_f8" -o trace -- python "$REPO/bench.py" --model detikzify-cl-7b --weight-format fp8 --steps 1 --warmup 0 --new-tokens 64 --no-cpu-baseline --batch 0 --probe-tokens 2 > "$OUT/prof_f8.log" 2>&1
cd "$REPO"; python tools/prof_summary.py "$OUT/prof_f8/trace_results.db" "$OUT/f8_kernel_stats.csv" 2>/dev/null | head -6 | cut -c1-160
