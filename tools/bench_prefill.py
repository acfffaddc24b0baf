#!/usr/bin/env python
"""ms of ViT + projector + prefill of the 243-token image prompt with the prefill's rows >> d GEMMs on k_gemm_mfma / k_gemm_px.
    python tools/bench_prefill.py [--model detikzify-ds-7b]"""
import argparse, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from detikzify_amd.model import load
from tests.helpers import sketch_image

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="detikzify-ds-7b")
args = ap.parse_args()
model, proc = load(args.model, synthetic=1234, batch_slots=2)
enc = proc(images=sketch_image(0, 224), return_tensors="pt")
ids, px = enc.input_ids[0], enc.pixel_values
for mode in (0, 1, 0, 1):
    model.set_option("gemm_px", mode)
    model.prefill(ids, px, slot=0)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); model.prefill(ids, px, slot=0, reuse=False); ts.append(time.perf_counter() - t0)
    st = model.stats()
    print(f"gemm_px {mode}: ViT + projector + {ids.numel()}-token prefill {1e3 * min(ts):6.2f} ms (ViT {st.get('last_vit_ms', 0):.2f} ms)", flush=True)
