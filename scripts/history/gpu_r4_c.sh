#!/usr/bin/env bash
# round 4, lease C: k_gemm_g3 (bit-identity test, ViT timing), hidden-visibility build smoke, the multi-vector tests on the
# measured block shapes, engine run length 1 vs 8, the persistent-layer probe at the ds-1.3b shape, and the first complete run
# of the new full-size parity tests (few-slot contexts, fp8 phases with the tie rule fixed, margin sign test, sampled peaked test).
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
echo "== smoke + GEMM identity + multi-vector tests"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_mv.py -q -p no:cacheprovider -k "three_stage or multi_vector or op_gemm" 2>&1 | tail -6
echo "== ViT per GEMM variant"
timeout 600 python tools/bench_vit.py > "$OUT/r04_bench_vit.txt" 2>&1; grep -E "batched|images per call" "$OUT/r04_bench_vit.txt" | grep -E "batched| 1 images| 8 images|16 images"
echo "== persistent-layer probe, ds-1.3b shape (d 2048, ff 5504)"
( cd tools/probe && timeout 180 ./engine2_probe_ds13b 8 20 ) > "$OUT/r04_engine2_probe_ds13b.txt" 2>&1; grep -E "us/layer|differ|CUs" "$OUT/r04_engine2_probe_ds13b.txt" | head -14
echo "== engine: per-step loop vs runs of 8 (64 trees x 2 expansions, stub reward)"
for rs in 1 8; do
  DTK_ENGINE_RUN_STEPS=$rs timeout 600 python bench.py --steps 1 --warmup 0 --skip-batched --mcts-trees 64 --no-config4 --no-config5 --no-cpu-baseline --mcts-seq-expansions 0 --mcts-oversubscribe 1 > "$OUT/r04_bench_engine_run_steps_$rs.json" 2> "$OUT/bench_rs$rs.err"
  python - "$OUT/r04_bench_engine_run_steps_$rs.json" $rs <<'EOF'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
p = d["mcts"]["parallel"]
print(f"run_steps {sys.argv[2]}: {p['rollouts_per_sec']:.2f} rollouts/s, {p['seconds']:.2f} s, engine {p['engine']}")
EOF
done
echo "== full-size parity (new tests)"
timeout 2400 python -m pytest tests/test_gpu_parity_batched.py -q -p no:cacheprovider -s --durations=8 -k "few_slot or (headline and fp8) or margins or peaked" 2>&1 | grep -v "^$" | tail -30 | cut -c1-1500 > "$OUT/r04c_parity.txt"; cat "$OUT/r04c_parity.txt"
