#!/usr/bin/env python
"""ViT and prefill time (device events, dtk_get_stats) under every GEMM implementation / tile / ring setting, one model load.
    python tools/tune_gemm.py --model detikzify-ds-7b
(Some of the variants timed here — k_gemm_b, k_gemv_bk, k_gemm_dma, the experiment modes of k_gemv_b — are only built with
DTK_EXPERIMENTS=1 ./build.sh; a default build reports "built without DTK_EXPERIMENTS" for them.)"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from detikzify_amd.model import load  # noqa: E402
from detikzify_amd.util.synthetic import sketch_image  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="detikzify-ds-7b")
args = ap.parse_args()
model, proc = load(args.model, synthetic=1234)
enc = proc(images=sketch_image(0, 224), return_tensors="pt")
ids, px = enc.input_ids[0], enc.pixel_values


def measure():
    best_v, best_p = 1e9, 1e9
    for _ in range(4):
        model.prefill(ids, px)
        st = model.stats()
        best_v, best_p = min(best_v, st["last_vit_ms"]), min(best_p, st["last_prefill_ms"])
    return best_v, best_p


rows = []
for impl, tile, ring in [(0, 0, 3)] + [(1, t, r) for t in (0, 1, 2, 3, 4) for r in (2, 3, 4)]:
    model.set_option("gemm_impl", impl); model.set_option("gemm_tile", tile); model.set_option("gemm_ring", ring)
    v, p = measure()
    name = {0: "auto", 1: "64x64", 2: "128x64", 3: "128x128", 4: "64x32"}[tile]
    print(f"impl {'dma ' if impl else 'regs'} tile {name:8s} ring {ring}: ViT {v:6.2f} ms   ViT + projector + {ids.numel()}-token prefill {p:6.2f} ms   (LLaMA part {p - v:6.2f})", flush=True)
