#!/usr/bin/env bash
# round 6, lease H: the two v2-8b test rules after their fix; 64 prefix groups (DTK_PFX_GROUPS 16 -> 64) against the step times of lease F;
# the ceiling of item 7 (ds-1.3b single-sequence step without its attention launch); the default bench line with the native engine and
# k_gemv_bc, and a DTK_TRACE_MCTS timeline of config 5.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
gcc -O2 -Iinclude tools/probe/step_bench.c -o tools/probe/step_bench -Ldetikzify_amd/lib -ldtk_hip -Wl,-rpath,"$REPO/detikzify_amd/lib" || exit 1
SB=$REPO/tools/probe/step_bench
{
echo "== cl-7b fp8, 64 slots, 8 images (lease F: 3.67 ms with 16 groups)"
STEP_BENCH_SLOTS=64 STEP_BENCH_IMAGES=8 STEP_BENCH_STEPS=48 timeout 600 $SB "" "gemv_bc=0" ""
echo "== ds-7b bf16, 64 slots, 1 image (lease F: 3.95-4.04)"
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=64 STEP_BENCH_STEPS=48 timeout 600 $SB "" "gemv_bc=0" ""
echo "== single sequence: the step with and without its attention launch (timing probe, wrong results)"
STEP_BENCH_MODEL=ds-1.3b STEP_BENCH_SLOTS=1 STEP_BENCH_STEPS=256 timeout 600 $SB "" "probe_skip_attn=1" ""
STEP_BENCH_MODEL=ds-7b STEP_BENCH_SLOTS=1 STEP_BENCH_STEPS=128 timeout 600 $SB "" "probe_skip_attn=1"
STEP_BENCH_SLOTS=1 STEP_BENCH_STEPS=128 timeout 600 $SB "" "probe_skip_attn=1"
} 2>&1 | sed -E 's/; last token.*//' | tee "$OUT/r06h_step_bench.txt"
timeout 900 python -m pytest tests -m gpu -x -q -k "headline_models_match_cpu_oracle and v2 or peaked_logits and (v2 or 1.3b) or batched_headline_matches_cpu_oracle and fp8-1" 2>&1 | tail -5 | tee "$OUT/r06h_pytest.txt"
timeout 1500 python bench.py --steps 20 --warmup 5 > "$OUT/r06h_bench_ds7b.json" 2> "$OUT/r06h_bench.err"; echo "bench exit $?"
C5="--no-cpu-baseline --skip-batched --mcts-trees 0 --mcts-seq-expansions 0 --no-config4 --no-rank-shapes --steps 1 --warmup 0 --probe-tokens 4"
DTK_TRACE_MCTS="$OUT/r06_mcts_trace.json" timeout 600 python bench.py $C5 > "$OUT/r06h_bench_config5_traced.json" 2>/dev/null; echo "config 5 traced: exit $?"
python tools/mcts_timeline.py "$OUT/r06_mcts_trace.json" > "$OUT/r06_mcts_timeline_config5.txt" 2>&1; rm -f "$OUT/r06_mcts_trace.json"; head -8 "$OUT/r06_mcts_timeline_config5.txt" | cut -c1-200
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r06h_bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    if "value" not in d: continue
    b = d.get("batched_rollouts") or {}; m = d.get("mcts") or {}
    c5 = m.get("config5") or {}
    g = lambda k: ((c5.get(k) or {}).get("rollouts_per_sec"))
    e = lambda k: ((c5.get(k) or {}).get("engine"))
    c4 = m.get("config4") or {}
    print(f.split("/")[-1], "tok/s", round(d["value"], 1), "decode", round(d.get("decode_tokens_per_sec_per_gpu") or 0, 1), "frac", round((d.get("decode_step") or {}).get("frac_of_hbm_peak") or 0, 3),
          "| batched", round(b.get("rollouts_per_sec", 0), 2), round(b.get("frac_of_hbm_peak", 0), 3), "| mcts seq", round((m.get("sequential") or {}).get("rollouts_per_sec", 0) or 0, 3),
          "par", round((m.get("parallel") or {}).get("rollouts_per_sec", 0) or 0, 2), "over", round((m.get("parallel_oversubscribed") or {}).get("rollouts_per_sec", 0) or 0, 2),
          "c4", (c4.get("fixed_length") or {}).get("rollouts_per_sec"), (c4.get("ragged") or {}).get("rollouts_per_sec"),
          "| c5 fixed", g("fixed_length"), "ragged", g("ragged"), "mx opt-in", g("fixed_length_fp8_matrix_cores_opt_in"),
          "| roofline", (d.get("roofline") or {}).get("frac"), "| cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("thread_choice"),
          "| secondary", d.get("secondary_rooflines"))
    print("   c5 fixed engine", e("fixed_length")); print("   c5 ragged engine", e("ragged"))
PY
