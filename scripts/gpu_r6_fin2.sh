#!/usr/bin/env bash
# round 6, lease FIN2 — the parts of lease FIN that the SwiGLU epilogue of the prefill's gate/up GEMM touches, on the final source: the default
# bench line (20 steps), cl-7b fp8 and v2-8b lines, the prefill per switch, kernel trace + FETCH_SIZE of a prefill, kernel trace + FETCH_SIZE + MFMA-busy of
# the bench (dominant_kernel.json / mfma_busy.json), config 5's timeline, the whole GPU suite, smoke()
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
R=${R:-r06fin2}
timeout 1500 python bench.py --steps 20 --warmup 5 > "$OUT/${R}_bench_ds7b.json" 2> "$OUT/${R}_bench.err"; echo "bench exit $?"
for cfg in "detikzify-cl-7b fp8" "detikzify-v2-8b bf16" "detikzify-ds-1.3b bf16"; do
  set -- $cfg
  timeout 900 python bench.py --model $1 --weight-format $2 --no-cpu-baseline --no-config5 --no-rank-shapes --steps 2 > "$OUT/${R}_bench_${1#detikzify-}_$2.json" 2>/dev/null; echo "$1 $2 exit $?"
done
{
SETS="prefill_sk=0,swiglu_fused=0;prefill_sk=1,swiglu_fused=0;prefill_sk=1,swiglu_fused=1;prefill_sk=1,swiglu_fused=1,qkv_rope_fused=0;prefill_sk=1,swiglu_fused=1,gemm_wt=0;prefill_sk=1,swiglu_fused=1"
timeout 600 python tools/bench_prefill.py --sets "$SETS" 2>&1 | grep -v "Warning\|amdgpu.ids"
timeout 600 python tools/bench_prefill.py --model detikzify-cl-7b --weight-format fp8 --sets "prefill_sk=0,swiglu_fused=0;prefill_sk=1,swiglu_fused=1" 2>&1 | grep -v "Warning\|amdgpu.ids"
timeout 600 python tools/bench_prefill.py --model detikzify-v2-8b --sets "prefill_sk=0,swiglu_fused=0;prefill_sk=1,swiglu_fused=1" 2>&1 | grep -v "Warning\|amdgpu.ids"
timeout 600 python tools/bench_prefill.py --model detikzify-ds-1.3b --sets "prefill_sk=0,swiglu_fused=0;prefill_sk=1,swiglu_fused=1" 2>&1 | grep -v "Warning\|amdgpu.ids"
} 2>&1 | tee "$OUT/${R}_prefill.txt"
C5="--no-cpu-baseline --skip-batched --mcts-trees 0 --mcts-seq-expansions 0 --no-config4 --no-rank-shapes --steps 1 --warmup 0 --probe-tokens 4"
DTK_TRACE_MCTS="$OUT/${R}_mcts_trace.json" timeout 600 python bench.py $C5 > "$OUT/${R}_bench_config5_traced.json" 2>/dev/null; echo "config 5 traced: exit $?"
python tools/mcts_timeline.py "$OUT/${R}_mcts_trace.json" > "$OUT/${R}_mcts_timeline_config5.txt" 2>&1; rm -f "$OUT/${R}_mcts_trace.json"; head -8 "$OUT/${R}_mcts_timeline_config5.txt" | cut -c1-200
SHORT="--steps 1 --warmup 0 --new-tokens 128 --no-cpu-baseline --batch 0 --mcts-seq-expansions 0 --no-config5 --probe-tokens 4"
cd /tmp && export TMPDIR=/tmp
prof() {   # name, counters ("" = kernel stats), command...; rocprofv3 on this image sometimes dies with a segmentation fault before the program starts: three tries
  local name=$1 ctrs=$2; shift 2
  for try in 1 2 3; do
    rm -rf "$OUT/prof_$name"
    if [ -z "$ctrs" ]; then timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
    else timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/prof_$name" -o pmc -- "$@" > "$OUT/prof_$name.log" 2>&1; fi
    local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/${R}_$name.csv" $([ -n "$ctrs" ] && echo --pmc) > /dev/null && break
  done
  rm -rf "$OUT/prof_$name" "$OUT/prof_$name.log"; echo "-- $name (try $try)"; head -8 "$OUT/${R}_$name.csv" | cut -c1-150
}
prof kernel_stats "" python "$REPO/bench.py" $SHORT
prof pmc_fetch "FETCH_SIZE" python "$REPO/bench.py" --steps 1 --warmup 0 --new-tokens 16 --no-cpu-baseline --batch 0 --mcts-seq-expansions 0 --no-config5 --probe-tokens 2
prof pmc_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" python "$REPO/bench.py" --steps 2 --warmup 0 --new-tokens 8 --no-cpu-baseline --batch 0 --mcts-seq-expansions 0 --no-config5 --probe-tokens 2
prof prefill_kernel_stats "" python "$REPO/tools/bench_prefill.py" --sets "prefill_sk=1" --reps 3 --rows 16
prof prefill_pmc_fetch "FETCH_SIZE" python "$REPO/tools/bench_prefill.py" --sets "prefill_sk=1" --reps 2 --rows 16
cd "$REPO"
python tools/make_dominant_kernel_json.py "$OUT/${R}_kernel_stats.csv" "$OUT/${R}_pmc_fetch.csv" detikzify-ds-7b > /dev/null && cp profiles/dominant_kernel.json "$OUT/dominant_kernel.json" && sed -i "s#gpurun_out/${R}_pmc_fetch.csv#profiles/${R}_pmc_fetch.csv#" "$OUT/dominant_kernel.json"
python tools/make_mfma_busy_json.py "$OUT/${R}_pmc_mfma.csv" detikzify-ds-7b > /dev/null && cp profiles/mfma_busy.json "$OUT/mfma_busy.json"
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/${R}_bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    if "value" not in d: continue
    b = d.get("batched_rollouts") or {}; m = d.get("mcts") or {}
    c5 = m.get("config5") or {}; c4 = m.get("config4") or {}
    g = lambda k: ((c5.get(k) or {}).get("rollouts_per_sec"))
    print(f.split("/")[-1], "tok/s", round(d["value"], 1), "decode", round(d.get("decode_tokens_per_sec_per_gpu") or 0, 1), "prefill_ms", round(d.get("prefill_ms") or 0, 2), "vit_ms", round(d.get("vit_ms") or 0, 2),
          "| batched", round(b.get("rollouts_per_sec", 0), 2), "| mcts seq", round((m.get("sequential") or {}).get("rollouts_per_sec", 0) or 0, 3),
          "par", round((m.get("parallel") or {}).get("rollouts_per_sec", 0) or 0, 2), "over", round((m.get("parallel_oversubscribed") or {}).get("rollouts_per_sec", 0) or 0, 2),
          "c4", (c4.get("fixed_length") or {}).get("rollouts_per_sec"), (c4.get("ragged") or {}).get("rollouts_per_sec"),
          "| c5 fixed", g("fixed_length"), "ragged", g("ragged"), "mx opt-in", g("fixed_length_fp8_matrix_cores_opt_in"),
          "| roofline", (d.get("roofline") or {}).get("frac"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
( time timeout 2400 python -m pytest tests -m gpu -q --durations=10 -rA ) 2>&1 | grep -v "^PASSED\|^SKIPPED" > "$OUT/${R}_pytest_gpu_full.txt"; tail -18 "$OUT/${R}_pytest_gpu_full.txt" | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/${R}_smoke.txt"
