#!/usr/bin/env bash
# round 2, call V: non-temporal KV loads in the batched attention kernel — attention time at private contexts, batched bench phase
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
for nt in 0 1 2; do
for cfg in "250 --private" "500 --private" "250 --fork"; do
  set -- $cfg
  DTK_OPTIONS="kv_nt=$nt" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_n" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 12 --ctx $1 $2 > "$OUT/prof_n.log" 2>&1
  python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_n -name trace_results.db | head -1)" "$OUT/prof_n.csv" > /dev/null 2>&1
  echo "kv_nt $nt ctx $1 $2: $(grep k_attn_tail_b "$OUT/prof_n.csv" | cut -d, -f1-6)"
  rm -rf "$OUT/prof_n"
done; done
cd "$REPO"
for nt in 0 1; do
  DTK_OPTIONS="kv_nt=$nt" timeout 600 python bench.py --steps 1 --warmup 1 --mcts-trees 0 --no-cpu-baseline > "$OUT/r2v_bench.log" 2> "$OUT/r2v_bench.err"
  python - "$OUT/r2v_bench.log" "kv_nt=$nt" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln); b = d["batched_rollouts"]
        print(sys.argv[2], "| batched rollouts/s", round(b["rollouts_per_sec"], 2), "frac", round(b["frac_of_hbm_peak"], 3), "ms/batch", round(b["ms_per_batch"]))
PY
done
