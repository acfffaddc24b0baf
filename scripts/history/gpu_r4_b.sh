#!/usr/bin/env bash
# round 4, lease B: fp8 batched step after the ring-of-8 k_gemv_br and the fp8 k_gemv_bkl + k_resid_norm_b path (identity tests, step
# times per option, kernel trace), bf16 qkv through k_gemv_br (experiment), the engine's multi-step runs against the per-step loop on
# the real device, and the first full-size runs of the new parity tests.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
echo "== identity tests of the batched kernels"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider --durations=5 -k "x_once_per_cu or 32_slot_batch or lds_staged" 2>&1 | tail -12
step() { echo "-- $1: $(DTK_OPTIONS="$1" timeout 300 python tools/bench_batch.py --batch 64 --fork --steps 32 ${@:2} 2>&1 | tail -1)"; }
echo "== fp8 64-slot step"
step "gemv_br_wd=8" --model detikzify-cl-7b --weight-format fp8
step "gemv_br_wd=4" --model detikzify-cl-7b --weight-format fp8
step "gemv_br_wd=8,resid_kparts=0" --model detikzify-cl-7b --weight-format fp8
step "gemv_br_wd=8,tail_threads=512" --model detikzify-cl-7b --weight-format fp8
echo "== bf16 64-slot step"
step "gemv_bl=33"
step "gemv_bl=97"
step "gemv_bl=33,tail_threads=512"
cd /tmp && export TMPDIR=/tmp
prof() {
  local name=$1; shift
  timeout 400 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o trace -- "$@" > "$OUT/prof_$name.log" 2>&1
  local db; db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r04_${name}_kernel_stats.csv" > /dev/null
  rm -rf "$OUT/prof_$name"
  echo "== $name: $(grep ms/step "$OUT/prof_$name.log")"; grep -E "gemv|attn|norm" "$OUT/r04_${name}_kernel_stats.csv" | head -9 | cut -c1-150
}
prof batch64_fp8_wd8 python "$REPO/tools/bench_batch.py" --batch 64 --fork --steps 16 --model detikzify-cl-7b --weight-format fp8
DTK_OPTIONS="gemv_bl=97" prof batch64_bf16_qkv_br python "$REPO/tools/bench_batch.py" --batch 64 --fork --steps 16
cd "$REPO"
echo "== engine: per-step loop vs multi-step runs (64 trees x 2 expansions, stub reward)"
for rs in 1 32; do
  DTK_ENGINE_RUN_STEPS=$rs timeout 600 python bench.py --steps 1 --warmup 0 --skip-batched --mcts-trees 64 --no-config4 --no-config5 --no-cpu-baseline --mcts-seq-expansions 0 --mcts-oversubscribe 1 > "$OUT/r04_bench_engine_run_steps_$rs.json" 2> "$OUT/bench_rs$rs.err"
  python - "$OUT/r04_bench_engine_run_steps_$rs.json" $rs <<'EOF'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
p = d["mcts"]["parallel"]
print(f"run_steps {sys.argv[2]}: {p['rollouts_per_sec']:.2f} rollouts/s, {p['seconds']:.2f} s, engine {p['engine']}")
EOF
done
echo "== full-size parity: few-slot contexts (ds-7b) and the fp8 65-slot phases"
timeout 1500 python -m pytest tests/test_gpu_parity_batched.py -q -x -p no:cacheprovider -s -k "few_slot and ds-7b or headline and cl-7b" 2>&1 | grep -v "^$" | tail -12
