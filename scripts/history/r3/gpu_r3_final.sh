#!/usr/bin/env bash
# round 3, the evidence run: rocprofv3 kernel traces + FETCH_SIZE passes (single-sequence step and the SHIPPED 64-slot step), the
# default bench line (CPU baselines, configs 4 / 5 inside), the other BASELINE model families, then the full GPU suite + smoke().
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
SHORT="--steps 1 --warmup 0 --new-tokens 128 --no-cpu-baseline --batch 0 --mcts-seq-expansions 0 --no-config5 --probe-tokens 4"
cd /tmp && export TMPDIR=/tmp
prof() {   # name, counters ("" = kernel stats), command...
  local name=$1 ctrs=$2; shift 2
  if [ -z "$ctrs" ]; then timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
  else timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/prof_$name" -o pmc -- "$@" > "$OUT/prof_$name.log" 2>&1; fi
  local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r03_$name.csv" $([ -n "$ctrs" ] && echo --pmc) > /dev/null
  rm -rf "$OUT/prof_$name"; head -8 "$OUT/r03_$name.csv" | cut -c1-150
}
prof kernel_stats "" python "$REPO/bench.py" $SHORT
prof pmc_fetch "FETCH_SIZE" python "$REPO/bench.py" --steps 1 --warmup 0 --new-tokens 16 --no-cpu-baseline --batch 0 --mcts-seq-expansions 0 --no-config5 --probe-tokens 2
prof batch64_kernel_stats "" python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork
prof batch64_pmc_fetch "FETCH_SIZE" python "$REPO/tools/bench_batch.py" --batch 64 --steps 8 --fork
prof batch64_ctx500_kernel_stats "" python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --ctx 500 --private
cd "$REPO"
python tools/make_dominant_kernel_json.py "$OUT/r03_kernel_stats.csv" "$OUT/r03_pmc_fetch.csv" detikzify-ds-7b && cp profiles/dominant_kernel.json "$OUT/dominant_kernel.json" && sed -i 's#gpurun_out/r03_pmc_fetch.csv#profiles/r03_pmc_fetch.csv#' profiles/dominant_kernel.json "$OUT/dominant_kernel.json"
timeout 1500 python bench.py > "$OUT/r03_bench_ds7b.json" 2> "$OUT/r3final_bench.err"; echo "bench exit $?"
for cfg in "detikzify-ds-1.3b bf16" "detikzify-cl-7b fp8" "detikzify-v2-8b bf16"; do
  set -- $cfg
  timeout 900 python bench.py --model $1 --weight-format $2 --no-cpu-baseline --no-config5 --steps 2 > "$OUT/r03_bench_${1#detikzify-}_$2.json" 2>/dev/null; echo "$1 $2 exit $?"
done
timeout 600 python bench.py --sample --no-cpu-baseline --batch 0 --mcts-seq-expansions 0 --no-config5 --steps 2 > "$OUT/r03_bench_ds7b_sampling.json" 2>/dev/null
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
timeout 1500 python -m pytest tests -m gpu -q --tb=short -s -p no:cacheprovider > "$OUT/r03_pytest_gpu.log" 2>&1
echo "pytest exit $?"
{ grep -E "passed|failed" "$OUT/r03_pytest_gpu.log" | tail -1; grep -E "^FAILED|^ERROR" "$OUT/r03_pytest_gpu.log"
  sed -E 's/^[.sFE]+//' "$OUT/r03_pytest_gpu.log" | grep -vE "^$|passed|failed|[Ww]arning|^  |^=|^-" | head -300; } > "$OUT/r03_pytest_gpu_summary.txt"
head -3 "$OUT/r03_pytest_gpu_summary.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
