#!/usr/bin/env bash
# round 5, lease I — after the closing run: the host-path and attention GPU tests on HEAD (the reward-prep pool became opt-in after
# lease H), smoke(), BASELINE config 3 (ds-7b, sampling decode, hipGraph per token), MFMA-busy counters (own --pmc pass) of the
# 8-image ViT pass and of the 64-slot step (the grouped prefix attention is the new MFMA kernel), ViT / prefill timing.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity_attn.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --tb=short -k "attn or engine or simulate or several_images or pipeline or kv_fork or resume or shared_prefix or batch" 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --sample --no-cpu-baseline --no-config4 --no-config5 --mcts-trees 0 --mcts-seq-expansions 0 --skip-batched --steps 5 --warmup 1 > "$OUT/r05_bench_ds7b_sampling.json" 2>/dev/null; echo "sampling bench exit $?"
python - <<PY
import json
d = json.loads([l for l in open("$OUT/r05_bench_ds7b_sampling.json") if l.startswith("{")][-1])
print("config 3 (sampling): tok/s", round(d["value"], 1), "decode", round(d.get("decode_tokens_per_sec_per_gpu") or 0, 1), "frac", round(d["decode_step"]["frac_of_hbm_peak"], 3), d["config"].get("workload", "")[:120])
PY
timeout 300 python tools/bench_vit.py 2>&1 | tail -25 > "$OUT/r05_bench_vit.txt"; head -12 "$OUT/r05_bench_vit.txt" | cut -c1-200
cd /tmp && export TMPDIR=/tmp
CTRS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
pmc() {   # name, command...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc $CTRS -d "$OUT/prof_$name" -o pmc -- "$@" > "$OUT/prof_$name.log" 2>&1
  local db=$(ls "$OUT"/prof_$name/*/*.db "$OUT"/prof_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r05_$name.csv" --pmc > /dev/null
  rm -rf "$OUT/prof_$name"; echo "-- $name"; head -14 "$OUT/r05_$name.csv" | cut -c1-170
}
pmc pmc_mfma_vit8 python "$REPO/tools/bench_vit.py" --only 8
STEP_BENCH_SLOTS=64 STEP_BENCH_MODEL=ds-7b STEP_BENCH_STEPS=12 pmc pmc_mfma_batch64 "$REPO/tools/probe/step_bench" ""
