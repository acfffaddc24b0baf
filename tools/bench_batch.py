#!/usr/bin/env python
"""Times the batched decode step (dtk_decode_batch_*) for B active slots at a given context length.
    python tools/bench_batch.py --model detikzify-ds-7b --batch 8 --steps 64"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from detikzify_amd.model import load  # noqa: E402
from detikzify_amd.util.synthetic import sketch_image  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="detikzify-ds-7b")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--graph", type=int, default=1)
ap.add_argument("--weight-format", default="bf16")
ap.add_argument("--slots", type=int, default=0, help="KV slots to allocate (default: --batch): partial-occupancy timing")
ap.add_argument("--ctx", type=int, default=0, help="text-only prompt of this many tokens instead of the image prompt (attention cost vs context)")
ap.add_argument("--private", action="store_true", help="with --ctx: every slot prefills its own prompt (no shared prefix)")
ap.add_argument("--max-positions", type=int, default=0, help="KV rows allocated per slot and head (default: the preset's): locality experiment")
ap.add_argument("--fork", action="store_true", help="prefill slot 0 only and fork its KV into the other slots (fewer dispatches: profiling runs)")
args = ap.parse_args()
model, proc = load(args.model, synthetic=1234, batch_slots=max(args.batch, args.slots), weight_format=args.weight_format,
                   max_positions=args.max_positions or None)
model.set_graph_mode(args.graph)
enc = proc(images=sketch_image(0, 224), return_tensors="pt")
ids, px = enc.input_ids[0], enc.pixel_values
if args.ctx:
    import torch
    ids, px = torch.randint(3, model.config.vocab - 1, (args.ctx,), generator=torch.Generator().manual_seed(1)), None
for s in range(args.batch):
    model.set_sampling(do_sample=False, bad_ids=[model.config.image_token_id], slot=s)
    if args.ctx and args.private:
        model.prefill((ids + s) % (model.config.vocab - 1), None, slot=s, reuse=False)
        continue
    if args.fork and s > 0:
        model.kv_fork(0, s, ids.numel())
    else:
        model.prefill(ids, px, slot=s, reuse=(s > 0))
slots = list(range(args.batch))
for _ in range(4):
    model.decode_batch_launch(slots); model.decode_batch_wait()
model.synchronize()
t0 = time.perf_counter()
model.decode_batch_launch(slots)
for _ in range(args.steps - 1):
    model.decode_batch_launch(slots)
    model.decode_batch_wait()
model.decode_batch_wait()
model.synchronize()
dt = time.perf_counter() - t0
st = model.stats()
ctx = ids.numel() + 4 + args.steps / 2
bytes_step = st["weight_bytes_per_token"] + args.batch * st["kv_bytes_per_ctx_token"] * ctx
print(f"B={args.batch}: {1e3 * dt / args.steps:.3f} ms/step, {args.batch * args.steps / dt:.1f} tok/s, "
      f"{bytes_step / (dt / args.steps) / 1e9:.0f} GB/s algorithmic")
