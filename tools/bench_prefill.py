#!/usr/bin/env python
"""Decoder prefill time (dtk_prefill's own HIP events: stats.last_prefill_ms - last_vit_ms) per setting of the sliced-K switches, for the
full image prompt and for short text-only prompts (the joins of a batch are such tails).
    python tools/bench_prefill.py [--model detikzify-ds-7b] [--weight-format bf16] [--sets "prefill_sk=0;prefill_sk=1;prefill_sk=4,gemm_sk_tile=1"]"""
import argparse, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from detikzify_amd.model import load
from detikzify_amd.util.image import expand
from detikzify_amd.util.synthetic import sketch_image

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="detikzify-ds-7b")
ap.add_argument("--weight-format", default="bf16")
ap.add_argument("--sets", default="prefill_sk=0;prefill_sk=1;prefill_sk=4;prefill_sk=2;prefill_sk=1,gemm_sk_tile=1;prefill_sk=1,gemm_sk_tile=0")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--rows", type=int, nargs="*", default=[16, 64, 128])
args = ap.parse_args()
model, proc = load(args.model, synthetic=1234, weight_format=args.weight_format)
model.reuse_prefix = False
img = sketch_image(0, 224); img = expand(img, max(img.size), do_trim=True)
enc = proc(images=img, return_tensors="pt")
ids, px = enc.input_ids, enc.pixel_values
g = torch.Generator().manual_seed(5)
text = {n: torch.randint(10, 1000, (1, n), generator=g) for n in args.rows}
first = None
for sset in args.sets.split(";"):
    for kv in filter(None, sset.split(",")):
        k, v = kv.split("="); model.set_option(k, int(v))
    out = []
    for name, (i, p) in [("image+prompt", (ids, px))] + [(f"{n} text rows", (text[n], None)) for n in args.rows]:
        ms = []
        for r in range(args.reps + 1):
            lg = model.prefill(i, p, return_logits=(r == 0), reuse=False)
            st = model.stats()
            ms.append(st["last_prefill_ms"] - st["last_vit_ms"])
        out.append(f"{name} ({i.shape[1]} rows): {min(ms[1:]):.3f} ms")
    print(f"{sset:40s} " + " | ".join(out), flush=True)
model.set_option("prefill_sk", 1); model.set_option("gemm_sk_tile", 2)
