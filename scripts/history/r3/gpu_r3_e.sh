#!/usr/bin/env bash
# round 3, lease E: two-phase weight ring of k_gemv_bkp + fp8 resid_split (identity tests, step time, rollouts/s), the MCTS timeline,
# counters of the batched ViT (MFMA busy, HBM fetch, L2 hit rate) under both GEMM kernels.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q --tb=short -s -p no:cacheprovider -k "x_once or fp8_weights or batched_decode or 32_slot" > "$OUT/r3e_tests.log" 2>&1
echo "tests exit $?"; tail -2 "$OUT/r3e_tests.log"
timeout 300 python tools/bench_batch.py --batch 64 --steps 24 --fork 2>&1 | tail -1
timeout 300 python tools/bench_batch.py --batch 64 --steps 24 --fork --model detikzify-cl-7b --weight-format fp8 2>&1 | tail -1 | sed "s/^/cl-7b fp8: /"
DTK_TRACE_MCTS=$OUT/r3e_trace.json timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-config5 --no-config4 --mcts-seq-expansions 0 > "$OUT/r3e_bench.json" 2> "$OUT/r3e_bench.err"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3e_bench.json") if l.startswith("{")][-1])
    b = d["batched_rollouts"]
    print("batched", round(b["rollouts_per_sec"], 2), "frac", round(b["frac_of_hbm_peak"], 3), "mcts parallel", d.get("mcts_rollouts_per_sec"), d["mcts"]["parallel"]["engine"])
except Exception as e:
    print("bench parse failed", repr(e))
PY
python tools/mcts_timeline.py "$OUT/r3e_trace.json" > "$OUT/r3e_mcts_timeline.txt" 2>&1; cat "$OUT/r3e_mcts_timeline.txt"
cd /tmp && export TMPDIR=/tmp
for impl in 0 2; do
  for pass in mfma fetch l2; do
    case $pass in
      mfma) CTRS="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE";;
      fetch) CTRS="FETCH_SIZE";;
      l2) CTRS="TCC_HIT_sum TCC_MISS_sum";;
    esac
    DTK_OPTIONS="gemm_impl=$impl" timeout 300 rocprofv3 --kernel-trace --pmc $CTRS -d "$OUT/pmc_vit_${impl}_$pass" -o pmc -- python "$REPO/tools/bench_vit.py" --only 8 > "$OUT/pmc_vit_${impl}_$pass.log" 2>&1
    db=$(ls "$OUT"/pmc_vit_${impl}_$pass/*/*.db "$OUT"/pmc_vit_${impl}_$pass/*.db 2>/dev/null | head -1)
    [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r3e_vit8_gemm${impl}_pmc_$pass.csv" --pmc | grep -i "gemm\|attention" | head -6
    rm -rf "$OUT/pmc_vit_${impl}_$pass"
  done
done
