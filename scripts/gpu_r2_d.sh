#!/usr/bin/env bash
# round 2, call D: parity of the LDS-staged batched GEMM + prefix test, batched sweep, per-kernel profile of two shapes
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider \
  -k "shared_prefix_on_matrix or 32_slot or lds_staged or batched_decode_tracks or batch_engine or kv_fork" > "$OUT/r2d_pytest.log" 2>&1
echo "pytest exit $?"; tail -6 "$OUT/r2d_pytest.log"
timeout 600 python tools/tune_batch.py --model detikzify-ds-7b --batch 64 --out "$OUT/tune_batch_ds7b_d.json" > "$OUT/tune_batch_ds7b_d.log" 2>&1
echo "tune batch exit $?"; grep -E "^---|ms/step" "$OUT/tune_batch_ds7b_d.log"
cd /tmp && export TMPDIR=/tmp
for shape in 1 2; do
  DTK_OPTIONS="attn_b_impl=1,prefix_mfma=0,tail_threads=256,gemm_b=$shape" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_gb$shape" -o trace -- python "$REPO/tools/bench_batch.py" --batch 64 --steps 24 --fork > "$OUT/prof_gb$shape.log" 2>&1
  python "$REPO/tools/prof_summary.py" "$(find $OUT/prof_gb$shape -name trace_results.db | head -1)" "$OUT/r02_batch_gemm${shape}_kernel_stats.csv" > /dev/null 2>&1
  rm -rf "$OUT/prof_gb$shape"; head -9 "$OUT/r02_batch_gemm${shape}_kernel_stats.csv" | cut -c1-140
done
