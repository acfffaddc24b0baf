#!/usr/bin/env python
"""Batched decode step (64 rollouts of one image) under every attention / GEMV-tile option, one model load:
ms per step at a short tail (context = image prefix + a few tokens) and at a long one (every slot has `--tail` private
tokens behind the shared prefix), graph replay.

    python tools/tune_batch.py --model detikzify-ds-7b --batch 64 --out gpurun_out/tune_batch_ds7b.json

(Some of the variants timed here — k_gemm_b, k_gemv_bk, k_gemm_dma, the experiment modes of k_gemv_b — are only built with
DTK_EXPERIMENTS=1 ./build.sh; a default build reports "built without DTK_EXPERIMENTS" for them.)"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from detikzify_amd.model import load  # noqa: E402
from detikzify_amd.util.synthetic import sketch_image  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="detikzify-ds-7b")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=24)
ap.add_argument("--tail", type=int, default=256)
ap.add_argument("--weight-format", default="bf16")
ap.add_argument("--out", default="gpurun_out/tune_batch.json")
args = ap.parse_args()
B = args.batch
model, proc = load(args.model, synthetic=1234, batch_slots=min(65, B + 1), weight_format=args.weight_format)
cfg = model.config
enc = proc(images=sketch_image(0, 224), return_tensors="pt")
ids, px = enc.input_ids[0], enc.pixel_values
slots = list(range(B))
st = model.stats()
W, Kb = st["weight_bytes_per_token"], st["kv_bytes_per_ctx_token"]


def setup(tail):
    """slot 0 holds the image prefix; every other slot forks it; with tail > 0 each slot then prefills its OWN random tokens"""
    g = torch.Generator().manual_seed(7)
    for s in slots:
        model.set_sampling(do_sample=True, temperature=0.8, top_p=0.95, seed=100 + s, bad_ids=[cfg.image_token_id],
                           always_suppress_ids=[cfg.eos_token_id], slot=s)
    model.prefill(ids, px, slot=0)
    for s in slots[1:]:
        model.kv_fork(0, s, ids.numel())
    if tail:
        for s in slots:
            extra = torch.randint(3, cfg.vocab - 1, (tail,), generator=g)
            extra[extra == cfg.image_token_id] = 3
            if s == 0:
                continue          # the source keeps the bare prefix (its rows are what the others share)
            model.prefill(torch.cat([ids, extra]), px, slot=s, reuse=True)


def ms_per_step(active):
    for _ in range(3):
        model.decode_batch_launch(active); model.decode_batch_wait()
    model.synchronize()
    t0 = time.perf_counter()
    model.decode_batch_launch(active)
    for _ in range(args.steps - 1):
        model.decode_batch_launch(active)
        model.decode_batch_wait()
    model.decode_batch_wait()
    model.synchronize()
    return 1e3 * (time.perf_counter() - t0) / args.steps


configs = [("tail kernel 256, no prefix kernel", dict(prefix_mfma=0, tail_threads=256)),
           ("prefix on MFMA 4 splits, tail 256", dict(prefix_mfma=1, pfx_splits=4)),
           ("x via LDS-DMA, 2x4 sk4 xa3 / 4x2 (1)", dict(prefix_mfma=0, gemm_b=1)),
           ("x via LDS-DMA, 4x2 sk2 xa3 / 4x2 (2)", dict(gemm_b=2)),
           ("x via LDS-DMA, 2x4 sk4 xa2 / 4x2 xa2 (3)", dict(gemm_b=3)),
           ("x via LDS-DMA, 2x4 sk2 xa3 / 8x1 (4)", dict(gemm_b=4)),
           ("shape 1 + prefix on MFMA", dict(gemm_b=1, prefix_mfma=1, pfx_splits=4)),
           ("shape 2 + prefix on MFMA", dict(gemm_b=2, prefix_mfma=1, pfx_splits=4))]
res = {"model": args.model, "batch": B, "rows": []}
for tail in (0, args.tail):
    ctx = ids.numel() + tail + 3 + args.steps / 2
    algo = W + B * Kb * ctx
    print(f"--- {args.model} B={B}, private tail {tail} tokens (context ~{ctx:.0f}); algorithmic {algo / 1e9:.2f} GB / step", flush=True)
    for name, opts in configs:
        for k, v in opts.items():
            model.set_option(k, v)
        setup(tail)
        active = slots if tail == 0 else slots[1:]       # long-tail run: the bare source does not decode
        ms = ms_per_step(active)
        n = len(active)
        a = W + n * Kb * ctx
        print(f"  {name:44s} {ms:7.3f} ms/step  {n * 1e3 / ms:8.0f} tok/s  {a / ms / 1e6:6.0f} GB/s = {a / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
        res["rows"].append({"tail": tail, "config": name, "options": opts, "ms_per_step": ms, "active": n, "frac_of_hbm_peak": a / ms / 1e6 / 8000})
Path(args.out).parent.mkdir(parents=True, exist_ok=True)
Path(args.out).write_text(json.dumps(res, indent=1))
