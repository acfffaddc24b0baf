#!/usr/bin/env bash
# round 5, lease F — the evidence run on the final source: the default bench line (headline + batched + MCTS + configs 4 / 5 + roofline +
# CPU baseline), the other BASELINE model families, rocprofv3 kernel trace + FETCH_SIZE pass of the single-sequence step (same command
# family as the bench line), kernel traces of the 64-slot steps, BASELINE config 5 with the reward-prep worker pool off (A/B) and a
# DTK_TRACE_MCTS timeline of it with the pool on.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 1500 python bench.py --steps 20 --warmup 5 > "$OUT/r05_bench_ds7b.json" 2> "$OUT/r05_bench.err"; echo "bench exit $?"
for cfg in "detikzify-ds-1.3b bf16" "detikzify-cl-7b fp8" "detikzify-v2-8b bf16"; do
  set -- $cfg
  timeout 900 python bench.py --model $1 --weight-format $2 --no-cpu-baseline --no-config5 --no-rank-shapes --steps 2 > "$OUT/r05_bench_${1#detikzify-}_$2.json" 2>/dev/null; echo "$1 $2 exit $?"
done
C5="--no-cpu-baseline --skip-batched --mcts-trees 0 --mcts-seq-expansions 0 --no-config4 --no-rank-shapes --steps 1 --warmup 0 --probe-tokens 4"
DTK_REWARD_PREP_WORKERS=0 timeout 600 python bench.py $C5 > "$OUT/r05_bench_config5_prep_pool_off.json" 2>/dev/null; echo "config 5, pool off: exit $?"
DTK_TRACE_MCTS="$OUT/r05_mcts_trace.json" timeout 600 python bench.py $C5 > "$OUT/r05_bench_config5_prep_pool_on_traced.json" 2>/dev/null; echo "config 5, pool on (traced): exit $?"
python tools/mcts_timeline.py "$OUT/r05_mcts_trace.json" > "$OUT/r05_mcts_timeline_config5.txt" 2>&1; rm -f "$OUT/r05_mcts_trace.json"; head -8 "$OUT/r05_mcts_timeline_config5.txt" | cut -c1-200
SHORT="--steps 1 --warmup 0 --new-tokens 128 --no-cpu-baseline --batch 0 --mcts-seq-expansions 0 --no-config5 --probe-tokens 4"
SB=$REPO/tools/probe/step_bench
cd /tmp && export TMPDIR=/tmp
prof() {   # name, counters ("" = kernel stats), command...
  local name=$1 ctrs=$2; shift 2
  if [ -z "$ctrs" ]; then timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
  else timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/prof_$name" -o pmc -- "$@" > "$OUT/prof_$name.log" 2>&1; fi
  local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r05_$name.csv" $([ -n "$ctrs" ] && echo --pmc) > /dev/null
  rm -rf "$OUT/prof_$name"; echo "-- $name"; head -8 "$OUT/r05_$name.csv" | cut -c1-150
}
prof kernel_stats "" python "$REPO/bench.py" $SHORT
prof pmc_fetch "FETCH_SIZE" python "$REPO/bench.py" --steps 1 --warmup 0 --new-tokens 16 --no-cpu-baseline --batch 0 --mcts-seq-expansions 0 --no-config5 --probe-tokens 2
STEP_BENCH_SLOTS=64 prof batch64_fp8_kernel_stats "" $SB ""
STEP_BENCH_SLOTS=64 prof batch64_fp8_mx_opt_in_kernel_stats "" $SB "act_fp8=1"
STEP_BENCH_SLOTS=64 STEP_BENCH_MODEL=ds-7b prof batch64_kernel_stats "" $SB ""
STEP_BENCH_SLOTS=64 prof batch64_fp8_pmc_fetch "FETCH_SIZE" $SB ""
cd "$REPO"
python tools/make_dominant_kernel_json.py "$OUT/r05_kernel_stats.csv" "$OUT/r05_pmc_fetch.csv" detikzify-ds-7b > /dev/null && cp profiles/dominant_kernel.json "$OUT/dominant_kernel.json" && sed -i 's#gpurun_out/r05_pmc_fetch.csv#profiles/r05_pmc_fetch.csv#' profiles/dominant_kernel.json "$OUT/dominant_kernel.json"
timeout 600 python -m pytest tests/test_gpu_parity.py::test_engine_prefix_paths_give_the_same_tokens_as_plain_generate tests/test_gpu_parity_attn.py -q -p no:cacheprovider --tb=short 2>&1 | tail -3
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r05_bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    if "value" not in d: continue
    b = d.get("batched_rollouts") or {}; m = d.get("mcts") or {}
    c5 = m.get("config5") or {}
    g = lambda k: ((c5.get(k) or {}).get("rollouts_per_sec"))
    c4 = ((m.get("config4") or {}).get("fixed_length") or {}).get("rollouts_per_sec")
    print(f.split("/")[-1], "tok/s", round(d["value"], 1), "decode", round(d.get("decode_tokens_per_sec_per_gpu") or 0, 1), "frac", round((d.get("decode_step") or {}).get("frac_of_hbm_peak") or 0, 3),
          "| batched", round(b.get("rollouts_per_sec", 0), 2), round(b.get("frac_of_hbm_peak", 0), 3), "| mcts seq", round((m.get("sequential") or {}).get("rollouts_per_sec", 0) or 0, 3),
          "par", round((m.get("parallel") or {}).get("rollouts_per_sec", 0) or 0, 2), "over", round((m.get("parallel_oversubscribed") or {}).get("rollouts_per_sec", 0) or 0, 2), "c4", c4,
          "| c5 fixed", g("fixed_length"), "ragged", g("ragged"), "mx opt-in", g("fixed_length_fp8_matrix_cores_opt_in"), "prep workers", m.get("reward_prep_workers"),
          "| roofline", (d.get("roofline") or {}).get("frac"), "| cpu", (d.get("cpu_baseline") or {}).get("value"),
          "| predicted c4", d.get("mcts_config4_predicted_rollouts_per_sec"), "c5", d.get("mcts_config5_predicted_rollouts_per_sec"))
PY
