#!/usr/bin/env bash
# round 4, lease E: the GEMM epilogues with their loads hoisted (all kernels) — identity tests, ViT per variant, prefill time,
# sampler identity after the un-spill, kernel trace of the 8-image ViT pass with k_gemm_g3.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "op_gemm or op_sample or sampl or vit or prefill_logits or multiblock" 2>&1 | tail -5
timeout 600 python tools/bench_vit.py > "$OUT/r04_bench_vit_epilogue.txt" 2>&1; grep -E "batched|images per call" "$OUT/r04_bench_vit_epilogue.txt" | grep -E "batched| 1 images| 8 images"
for impl in 3 4; do
DTK_OPTIONS="gemm_impl=$impl" timeout 300 python - <<'EOF'
import os, sys, time
sys.path.insert(0, os.getcwd())
from detikzify_amd.model import load
from detikzify_amd.util.synthetic import sketch_image
model, proc = load("detikzify-ds-7b", synthetic=1234)
enc = proc(images=sketch_image(0, 224), return_tensors="pt")
ids, px = enc.input_ids[0], enc.pixel_values
for _ in range(3):
    model.prefill(ids, px)
st = model.stats()
print(f"gemm_impl {os.environ['DTK_OPTIONS']}: ViT + projector + prefill {st['last_prefill_ms']:.2f} ms (ViT {st['last_vit_ms']:.2f} ms)")
EOF
done
cd /tmp && export TMPDIR=/tmp
DTK_OPTIONS="gemm_impl=4" timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_v" -o trace -- python "$REPO/tools/bench_vit.py" --only 8 > "$OUT/prof_v.log" 2>&1
db=$(ls "$OUT"/prof_v/*/*.db "$OUT"/prof_v/*.db 2>/dev/null | head -1); [ -n "$db" ] && python "$REPO/tools/prof_summary.py" "$db" "$OUT/r04_vit8_g3_epilogue_kernel_stats.csv" > /dev/null; rm -rf "$OUT/prof_v"
grep -E "gemm|attention|layernorm" "$OUT/r04_vit8_g3_epilogue_kernel_stats.csv" | head -6 | cut -c1-150
