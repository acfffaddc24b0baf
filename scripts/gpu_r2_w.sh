#!/usr/bin/env bash
# round 2, call W: the evidence run on the final tree (same commands as call G) — rocprofv3 kernel trace + FETCH_SIZE pass of the default bench command, batched-step trace,
# the default bench line (with CPU baselines) and the other BASELINE configurations
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
SHORT="--steps 1 --warmup 0 --new-tokens 128 --no-cpu-baseline --batch 0 --mcts-seq-expansions 0 --probe-tokens 4"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_g" -o trace -- python "$REPO/bench.py" $SHORT > "$OUT/r2w_prof.log" 2>&1; echo "rocprof exit $?"
python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_g -name trace_results.db | head -1)" "$OUT/r02_kernel_stats.csv" > /dev/null; rm -rf "$OUT/prof_g"; head -12 "$OUT/r02_kernel_stats.csv" | cut -c1-130
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/prof_gp" -o pmc -- python "$REPO/bench.py" --steps 1 --warmup 0 --new-tokens 16 --no-cpu-baseline --batch 0 --mcts-seq-expansions 0 --probe-tokens 2 > "$OUT/r2w_pmc.log" 2>&1; echo "rocprof pmc exit $?"
python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_gp -name 'pmc_results.db' | head -1)" "$OUT/r02_pmc_fetch.csv" --pmc > /dev/null; rm -rf "$OUT/prof_gp"; head -8 "$OUT/r02_pmc_fetch.csv" | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_gb" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork > "$OUT/r2w_prof_batch.log" 2>&1
python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_gb -name trace_results.db | head -1)" "$OUT/r02_batch64_kernel_stats.csv" > /dev/null; rm -rf "$OUT/prof_gb"; head -9 "$OUT/r02_batch64_kernel_stats.csv" | cut -c1-130; grep "ms/step" "$OUT/r2w_prof_batch.log"
cd "$REPO"
timeout 1500 python bench.py > "$OUT/r02_bench_ds7b.json" 2> "$OUT/r2w_bench.err"; echo "bench exit $?"
for cfg in "detikzify-ds-1.3b bf16" "detikzify-cl-7b fp8" "detikzify-v2-8b bf16"; do
  set -- $cfg
  timeout 900 python bench.py --model $1 --weight-format $2 --no-cpu-baseline --steps 2 > "$OUT/r02_bench_${1#detikzify-}_$2.json" 2>/dev/null; echo "$1 $2 exit $?"
done
timeout 600 python bench.py --sample --no-cpu-baseline --batch 0 --mcts-seq-expansions 0 --steps 2 > "$OUT/r02_bench_ds7b_sampling.json" 2>/dev/null
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > "$OUT/r2w_pytest.log" 2>&1; echo "pytest exit $?"; tail -3 "$OUT/r2w_pytest.log"
grep -E "ViT per-block|decoder depth|resume in place|dev-vs-bf16-oracle|passed|failed" "$OUT/r2w_pytest.log" | cut -c1-600 > "$OUT/r02_pytest_gpu_summary.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> "$OUT/r02_pytest_gpu_summary.txt"
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r02_bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    b = d.get("batched_rollouts") or {}; m = d.get("mcts") or {}
    print(f.split("/")[-1], "tok/s", round(d["value"], 1), "decode", round(d["decode_tokens_per_sec_per_gpu"], 1), "frac", round(d["decode_step"]["frac_of_hbm_peak"], 3),
          "| batched", round(b.get("rollouts_per_sec", 0), 2), round(b.get("frac_of_hbm_peak", 0), 3), "| mcts seq", round((m.get("sequential") or {}).get("rollouts_per_sec", 0), 3),
          "par", round((m.get("parallel") or {}).get("rollouts_per_sec", 0), 2), "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"),
          (d.get("cpu_baseline") or {}).get("parity_tokens_identical"))
PY
