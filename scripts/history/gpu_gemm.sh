#!/usr/bin/env bash
# GEMM tile x register-stage sweep on the ViT and the ds-7b prefill
set -uo pipefail
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider --tb=short -k "gemm or vit or prefill or greedy or long_context" 2>&1 | tail -3
timeout 900 python - <<'PY' 2>&1 | tail -6
import time, torch
from detikzify_amd.model import load
from tests.helpers import sketch_image
model, proc = load("detikzify-ds-7b", synthetic=7)
enc = proc(images=sketch_image(3, 384), return_tensors="pt")
ids = enc.input_ids[0]
for tile in (64, 128):
    for st in (2, 3):
        model.set_option("gemm_bk", tile); model.set_option("gemm_stages", st)
        for rep in range(2):
            model.synchronize(); t0 = time.perf_counter()
            for _ in range(6): model.vit_encode(enc.pixel_values)
            model.synchronize(); tv = (time.perf_counter() - t0) / 6
        for i in range(4): model.prefill(ids, enc.pixel_values, reuse=False)
        s = model.stats()
        print(f"bk={tile} stages={st}: vit_encode {tv*1e3:.2f} ms   device prefill {s['last_prefill_ms']:.2f} ms (vit {s['last_vit_ms']:.2f})")
PY
