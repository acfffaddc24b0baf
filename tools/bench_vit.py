#!/usr/bin/env python
"""ms per image of the vision tower (dtk_vit_encode, pooled output: the SelfSim reward's call) for 1..16 images per call, per GEMM
tile, and whether a batched image equals the same image encoded alone.   python tools/bench_vit.py [--model detikzify-ds-7b]"""
import argparse, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from detikzify_amd.model import load
from detikzify_amd.util.synthetic import sketch_image

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="detikzify-ds-7b")
ap.add_argument("--only", type=int, default=0, help="profiling runs: this many images per call only, GEMM variant from DTK_OPTIONS")
args = ap.parse_args()
model, proc = load(args.model, synthetic=1234)
px = torch.cat([proc(images=sketch_image(i, 224), return_tensors="pt").pixel_values for i in range(16)])
if args.only:
    for _ in range(6):
        model.vit_encode(px[:args.only], want_pooled=True, want_feats=False)
    sys.exit(0)
alone = [model.vit_encode(px[i:i + 1], want_pooled=True) for i in range(3)]
f8, p8 = model.vit_encode(px[:8], want_pooled=True)
print("batched == alone (features, pooled):", all(torch.equal(f8[i], alone[i][0][0]) for i in range(3)), all(torch.equal(p8[i], alone[i][1][0]) for i in range(3)))
ap_variants = ((0, "auto", 3), (1, "64x64", 0), (0, "glds128", 2), (0, "g3", 4))   # (gemm_tile, name, gemm_impl)
for tile, name, impl in ap_variants:
    model.set_option("gemm_tile", tile); model.set_option("gemm_impl", impl)
    for B in (1, 2, 4, 8, 16):
        model.vit_encode(px[:B], want_pooled=True, want_feats=False)
        t0 = time.perf_counter(); n = 4
        for _ in range(n):
            model.vit_encode(px[:B], want_pooled=True, want_feats=False)
        dt = (time.perf_counter() - t0) / n
        print(f"tile {name:8s} {B:2d} images per call: {1e3 * dt / B:6.2f} ms per image ({666.5e-3 * B / dt:6.0f} TFLOP/s)", flush=True)
model.set_option("gemm_tile", 0); model.set_option("gemm_impl", 3)
