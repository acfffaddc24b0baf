#!/usr/bin/env bash
# round 4, the closing evidence run on the final source (lease G ran the full GPU suite + the single-sequence traces earlier in the
# round; leases H-J brought up the fp8 matrix-core step): the GPU tests the later changes touch, smoke(), the default bench line
# (config 5 in both activation modes), the other BASELINE model families, kernel traces of the 64-slot step (bf16 model; fp8 model on
# the fp8 matrix cores) and a FETCH_SIZE pass of the latter.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -s --tb=short -k "(x_once_per_cu and cl-7b) or (few_slot and cl-7b) or fp8_weights_parity or test_mx_" 2>&1 | grep -vE "amdgpu.ids|^$" | cut -c1-1500 | tail -40 | tee "$OUT/r04_final_tests.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py --steps 20 --warmup 5 > "$OUT/r04_bench_ds7b.json" 2> "$OUT/r4final_bench.err"; echo "bench exit $?"
for cfg in "detikzify-ds-1.3b bf16" "detikzify-cl-7b fp8" "detikzify-v2-8b bf16"; do
  set -- $cfg
  timeout 900 python bench.py --model $1 --weight-format $2 --no-cpu-baseline --no-config5 --no-rank-shapes --steps 2 > "$OUT/r04_bench_${1#detikzify-}_$2.json" 2>/dev/null; echo "$1 $2 exit $?"
done
cd /tmp && export TMPDIR=/tmp
prof() {   # name, counters ("" = kernel stats), command...
  local name=$1 ctrs=$2; shift 2
  if [ -z "$ctrs" ]; then timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
  else timeout 600 rocprofv3 --kernel-trace --pmc $ctrs -d "$OUT/prof_$name" -o pmc -- "$@" > "$OUT/prof_$name.log" 2>&1; fi
  local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r04_$name.csv" $([ -n "$ctrs" ] && echo --pmc) > /dev/null
  rm -rf "$OUT/prof_$name"; head -8 "$OUT/r04_$name.csv" | cut -c1-150
}
prof batch64_kernel_stats "" python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork
prof batch64_fp8_mx_kernel_stats "" python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork --model detikzify-cl-7b --weight-format fp8
prof batch64_fp8_mx_pmc_fetch "FETCH_SIZE" python "$REPO/tools/bench_batch.py" --batch 64 --steps 8 --fork --model detikzify-cl-7b --weight-format fp8
cd "$REPO"
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/r04_bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    if "value" not in d: continue
    b = d.get("batched_rollouts") or {}; m = d.get("mcts") or {}
    c5 = m.get("config5") or {}
    g = lambda k: ((c5.get(k) or {}).get("rollouts_per_sec"))
    c4 = ((m.get("config4") or {}).get("fixed_length") or {}).get("rollouts_per_sec")
    print(f.split("/")[-1], "tok/s", round(d["value"], 1), "decode", round(d["decode_tokens_per_sec_per_gpu"], 1), "frac", round(d["decode_step"]["frac_of_hbm_peak"], 3),
          "| batched", round(b.get("rollouts_per_sec", 0), 2), round(b.get("frac_of_hbm_peak", 0), 3), "| mcts seq", round((m.get("sequential") or {}).get("rollouts_per_sec", 0) or 0, 3),
          "par", round((m.get("parallel") or {}).get("rollouts_per_sec", 0) or 0, 2), "c4", c4, "| c5 fixed", g("fixed_length"), "ragged", g("ragged"), "bf16-act", g("fixed_length_bf16_activations"),
          "on fp8 mfma", c5.get("decode_steps_on_fp8_matrix_cores"), "| roofline", (d.get("roofline") or {}).get("frac"), "| cpu", (d.get("cpu_baseline") or {}).get("value"),
          "| predicted c4", d.get("mcts_config4_predicted_rollouts_per_sec"), "c5", d.get("mcts_config5_predicted_rollouts_per_sec"))
PY
