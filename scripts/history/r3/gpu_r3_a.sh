#!/usr/bin/env bash
# round 3, lease A: the new full-size parity tests of the batched (rollouts/sec) path + long contexts, the tests touched by this
# round's C-side changes (sampler snapshot, 72 slots), the default bench line (config4 / config5 inside) and the f3 measurement.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests/test_gpu_parity_batched.py -q --tb=short -s -p no:cacheprovider > "$OUT/r3a_batched_parity.log" 2>&1
echo "batched parity exit $?"; tail -3 "$OUT/r3a_batched_parity.log"
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -s -p no:cacheprovider -k "multiblock or resume or simulate_parallel or several_images or batch_engine or kv_fork or shared_prefix_reads or v2_batched or 32_slot" > "$OUT/r3a_subset.log" 2>&1
echo "subset exit $?"; tail -2 "$OUT/r3a_subset.log"
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -s -p no:cacheprovider -k "x_once_per_cu" > "$OUT/r3a_kparts_identity.log" 2>&1
echo "kparts identity exit $?"; tail -2 "$OUT/r3a_kparts_identity.log"
# batched phase only, new N = d kernels on: A/B against the default line below
DTK_OPTIONS="resid_kparts=1" timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --mcts-seq-expansions 0 --mcts-trees 0 --no-config4 --no-config5 > "$OUT/r3a_bench_kparts.json" 2> "$OUT/r3a_bench_kparts.err"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3a_bench_kparts.json") if l.startswith("{")][-1])
    b = d["batched_rollouts"]
    print("kparts: batched", round(b["rollouts_per_sec"], 2), "frac", round(b["frac_of_hbm_peak"], 3), b["engine_seconds"])
except Exception as e:
    print("kparts bench parse failed", repr(e))
PY
timeout 900 python bench.py > "$OUT/r3a_bench.json" 2> "$OUT/r3a_bench.err"
echo "bench exit $?"; python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3a_bench.json") if l.startswith("{")][-1])
    b, m = d.get("batched_rollouts", {}), d.get("mcts", {})
    print("tok/s", round(d["value"], 1), "decode frac", round(d["decode_step"]["frac_of_hbm_peak"], 4), "roofline", round(d["roofline"]["frac"], 4))
    print("batched", round(b.get("rollouts_per_sec", 0), 2), "frac", round(b.get("frac_of_hbm_peak", 0), 3), "survey", round(b.get("frac_of_survey_formula", 0), 3))
    print("mcts seq/par", d.get("mcts_rollouts_per_sec_sequential"), d.get("mcts_rollouts_per_sec"))
    for k in ("config4", "config5"):
        c = m.get(k, {})
        print(k, {v: (round(c[v].get("rollouts_per_sec", 0), 2), c[v].get("seconds"), c[v].get("gather_seconds"), c[v].get("frac_of_roofline")) for v in ("fixed_length", "ragged") if v in c}, c.get("error"), c.get("model_load_seconds"))
    print("mcts error", m.get("error"))
except Exception as e:
    print("bench parse failed", repr(e))
PY
tail -5 "$OUT/r3a_bench.err"
timeout 600 python bench.py --steps 1 --warmup 0 --skip-batched --no-config4 --no-config5 --no-cpu-baseline --mcts-seq-expansions 0 --reward-latency 1 5 > "$OUT/r3a_bench_reward_latency.json" 2> "$OUT/r3a_bench_reward_latency.err"
echo "reward-latency exit $?"; python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3a_bench_reward_latency.json") if l.startswith("{")][-1])
    for S, e in d["mcts"].get("reward_latency", {}).items():
        for k, v in e.items():
            print(S, k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
    print("err", d["mcts"].get("error"))
except Exception as e:
    print("parse failed", repr(e))
PY
# per-kernel times of the 64-slot step, default vs resid_kparts (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
for name in default kparts; do
  opts=""; [ "$name" = kparts ] && opts="resid_kparts=1"
  DTK_OPTIONS="$opts" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_b64_$name" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork > "$OUT/prof_b64_$name.log" 2>&1
  db=$(ls "$OUT"/prof_b64_$name/*/*.db "$OUT"/prof_b64_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r3a_batch64_${name}_kernel_stats.csv" && head -12 "$OUT/r3a_batch64_${name}_kernel_stats.csv"
  rm -rf "$OUT/prof_b64_$name"
done
