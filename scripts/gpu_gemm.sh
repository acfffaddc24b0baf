#!/usr/bin/env bash
set -uo pipefail
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "gemm or vit or prefill or full_size or greedy" 2>&1 | tail -6
timeout 600 python - <<'PY' 2>&1 | tail -8
import time, torch
from detikzify_amd.model import load
from tests.helpers import sketch_image
model, proc = load("detikzify-ds-7b", synthetic=7)
enc = proc(images=sketch_image(3, 384), return_tensors="pt")
ids = enc.input_ids[0]
for tile in (1, 2, 3, 4):
    model.set_option("gemm_stages", tile)
    for rep in range(2):
        model.synchronize(); t0 = time.perf_counter()
        for _ in range(10): model.vit_encode(enc.pixel_values)
        model.synchronize(); tv = (time.perf_counter() - t0) / 10
    model.synchronize(); t0 = time.perf_counter()
    for i in range(5): model.prefill(ids, enc.pixel_values, reuse=False)
    model.synchronize(); tp = (time.perf_counter() - t0) / 5
    st = model.stats()
    print(f"gemm_stages={tile}: vit_encode {tv*1e3:.2f} ms   prefill {tp*1e3:.2f} ms (device: prefill {st['last_prefill_ms']:.2f}, vit {st['last_vit_ms']:.2f})")
PY
