#!/usr/bin/env bash
# MFMA flash attention: parity tests + ViT / prefill timing against the VALU kernel.
set -uo pipefail
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "attention or vit or prefill or loader or full_size or selfsim or greedy" 2>&1 | tail -15
timeout 600 python - <<'PY' 2>&1 | tail -12
import time, torch
from detikzify_amd.model import load
from tests.helpers import sketch_image
for name in ("detikzify-ds-7b",):
    model, proc = load(name, synthetic=7)
    enc = proc(images=sketch_image(3, 384), return_tensors="pt")
    for impl in (1, 2):
        model.set_option("attn_impl", impl)
        for rep in range(2):
            model.synchronize(); t0 = time.perf_counter()
            for _ in range(10): model.vit_encode(enc.pixel_values)
            model.synchronize(); tv = (time.perf_counter() - t0) / 10
        ids = enc.input_ids[0]
        model.synchronize(); t0 = time.perf_counter()
        for i in range(5): model.prefill(ids, enc.pixel_values, reuse=False)
        model.synchronize(); tp = (time.perf_counter() - t0) / 5
        print(f"{name} attn_impl={impl}: vit_encode {tv*1e3:.2f} ms   prefill(image+{len(ids)} tokens) {tp*1e3:.2f} ms")
PY
