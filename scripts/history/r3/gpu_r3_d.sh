#!/usr/bin/env bash
# round 3, lease D: ping-pong K/V prefetch in the batched attention + batched ViT attention: identity tests, step time, rollouts/s
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -s -p no:cacheprovider -k "batched_decode or kv_fork or shared_prefix or v2_batched or gqa or 32_slot or vit_batch or vit_features or x_once or op_attention or several_images or resume" > "$OUT/r3d_tests.log" 2>&1
echo "tests exit $?"; tail -3 "$OUT/r3d_tests.log"
for ctx in 0 500; do
  extra=""; [ "$ctx" != 0 ] && extra="--ctx 500 --private"
  timeout 300 python tools/bench_batch.py --batch 64 --steps 24 --fork $extra 2>&1 | tail -1 | sed "s/^/ctx $ctx: /"
done
timeout 300 python tools/bench_vit.py 2>&1 | grep -E "auto" 
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config5 --mcts-seq-expansions 0 > "$OUT/r3d_bench.json" 2> "$OUT/r3d_bench.err"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3d_bench.json") if l.startswith("{")][-1])
    b = d["batched_rollouts"]
    print("batched", round(b["rollouts_per_sec"], 2), "frac", round(b["frac_of_hbm_peak"], 3), "mcts parallel", d.get("mcts_rollouts_per_sec"), "config4", d.get("mcts_config4_rollouts_per_sec"))
except Exception as e:
    print("bench parse failed", repr(e))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_b64" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --ctx 500 --private > "$OUT/prof_b64.log" 2>&1
db=$(ls "$OUT"/prof_b64/*/*.db "$OUT"/prof_b64/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r3d_batch64_ctx500_kernel_stats.csv" >/dev/null && grep -v "fill_synth\|rocclr\|retile" "$OUT/r3d_batch64_ctx500_kernel_stats.csv" | head -9
rm -rf "$OUT/prof_b64"
