#!/usr/bin/env bash
# round 3, lease B: the persistent-layer probe (loader / consumer waves + granule hand-offs vs the launch chain), the new 128 x 128
# LDS-DMA GEMM (bit-identity tests, ViT and prefill timing), and the KV-locality experiment for the batched attention.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
( cd tools/probe && timeout 120 ./engine2_probe 8 20 ) > "$OUT/r3b_engine2_probe.txt" 2>&1
echo "engine probe exit $?"; cat "$OUT/r3b_engine2_probe.txt"
timeout 300 python -m pytest tests/test_gpu_parity.py -q --tb=short -s -p no:cacheprovider -k "op_gemm" > "$OUT/r3b_gemm_tests.log" 2>&1
echo "gemm tests exit $?"; tail -3 "$OUT/r3b_gemm_tests.log"
timeout 300 python tools/bench_vit.py > "$OUT/r3b_bench_vit.txt" 2>&1; grep -E "auto|glds|batched" "$OUT/r3b_bench_vit.txt"
for impl in 0 2; do
  DTK_OPTIONS="gemm_impl=$impl" timeout 300 python bench.py --steps 2 --warmup 1 --new-tokens 32 --batch 0 --no-cpu-baseline --mcts-seq-expansions 0 --no-config5 --probe-tokens 2 > "$OUT/r3b_bench_gemm$impl.json" 2> "$OUT/r3b_bench_gemm$impl.err"
  python - "$impl" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f"gpurun_out/r3b_bench_gemm{sys.argv[1]}.json") if l.startswith("{")][-1])
    print("gemm_impl", sys.argv[1], "prefill_ms", round(d["prefill_ms"], 2), "vit_ms", round(d["vit_ms"], 2))
except Exception as e:
    print("parse failed", repr(e))
PY
done
timeout 900 python -m pytest tests/test_gpu_parity_batched.py -q --tb=short -s -p no:cacheprovider -k "cl-7b or peaked" > "$OUT/r3b_parity_rerun.log" 2>&1
echo "parity rerun exit $?"; grep -E "^batched|^peaked|passed|failed|^E " "$OUT/r3b_parity_rerun.log" | head -12
timeout 400 python bench.py --steps 1 --warmup 0 --skip-batched --no-config4 --no-config5 --no-cpu-baseline --mcts-seq-expansions 0 --reward-latency 1 5 > "$OUT/r3b_bench_reward_latency.json" 2> "$OUT/r3b_bench_reward_latency.err"
echo "reward-latency exit $?"; python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3b_bench_reward_latency.json") if l.startswith("{")][-1])
    for S, e in d["mcts"].get("reward_latency", {}).items():
        for k, v in e.items():
            print(S, k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
    print("err", d["mcts"].get("error"), "parallel", d.get("mcts_rollouts_per_sec"))
except Exception as e:
    print("parse failed", repr(e))
PY
# batched attention vs the KV stride: 64 slots, 500 private keys each, KV rows allocated per (slot, head) = 2048 vs 640
for mp in 2048 640; do
  timeout 300 python tools/bench_batch.py --batch 64 --steps 24 --ctx 500 --private --max-positions $mp 2>&1 | tail -1 | sed "s/^/max_positions $mp: /"
done
