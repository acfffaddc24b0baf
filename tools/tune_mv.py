#!/usr/bin/env python
"""Times the multi-vector decode step (contexts with <= 5 slots, csrc/kernels_decode_mv.hip) for 1 / 2 / 4 active slots: first
the defaults next to the MFMA family on the same context (option mv_slots = 0) and the single-sequence graph, then every block
shape of every role with the others at their default, then the attention block sizes.
    python tools/tune_mv.py --model detikzify-ds-7b [--weight-format fp8] [--steps 64] [--quick]"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from detikzify_amd.model import load  # noqa: E402
from detikzify_amd.util.synthetic import sketch_image  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="detikzify-ds-7b")
ap.add_argument("--weight-format", default="bf16")
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--quick", action="store_true", help="defaults only")
args = ap.parse_args()

model, proc = load(args.model, synthetic=1234, batch_slots=5, weight_format=args.weight_format)
enc = proc(images=sketch_image(0, 224), return_tensors="pt")
ids, px = enc.input_ids[0], enc.pixel_values
img = model.config.image_token_id


def prepare(n):
    model.set_sampling(do_sample=False, bad_ids=[img], slot=4)
    model.prefill(ids, px, slot=4)
    for s in range(n):
        model.set_sampling(do_sample=True, temperature=0.8, top_p=0.95, seed=100 + s, bad_ids=[img], slot=s)
        model.kv_fork(4, s, ids.numel())


def step_ms(n, steps=None):
    steps = steps or args.steps
    prepare(n)
    slots = list(range(n))
    for _ in range(4):
        model.decode_batch_launch(slots); model.decode_batch_wait()
    model.synchronize()
    t0 = time.perf_counter()
    model.decode_batch_launch(slots)
    for _ in range(steps - 1):
        model.decode_batch_launch(slots)
        model.decode_batch_wait()
    model.decode_batch_wait()
    model.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def single_ms(steps=None):
    steps = steps or args.steps
    model.set_sampling(do_sample=True, temperature=0.8, top_p=0.95, seed=7, bad_ids=[img])
    model.prefill(ids, px)
    for _ in range(4):
        model.decode_launch(); model.decode_wait()
    model.synchronize()
    t0 = time.perf_counter()
    model.decode_launch()
    for _ in range(steps - 1):
        model.decode_launch(); model.decode_wait()
    model.decode_wait()
    model.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


print(f"# {args.model} {args.weight_format}: ms per step at the image-prefix context (243 + ~{args.steps // 2} keys), {args.steps} steps")
print(f"single-sequence graph: {single_ms():.3f} ms")
for n in (1, 2, 4):
    mv = step_ms(n)
    model.set_option("mv_slots", 0)
    mfma = step_ms(n)
    model.set_option("mv_slots", 4)
    print(f"{n} slots: multi-vector {mv:.3f} ms, MFMA family (one 16-column tile) {mfma:.3f} ms")
if not args.quick:
    for n in (2, 4):
        base = step_ms(n)
        for role in ("qkv", "o", "gu", "down", "lm_head"):
            row = []
            for shape in (0, 1, 2, 3):
                model.set_option(f"mv_shape_{role}", shape)
                row.append(step_ms(n, 32))
                model.set_option(f"mv_shape_{role}", -1)
            print(f"{n} slots, {role:8s}: default {base:.3f} | shapes 0..3: " + " ".join(f"{v:.3f}" for v in row))
        for threads in (256, 512, 1024):
            model.set_option("mv_tail_threads", threads)
            print(f"{n} slots, attention block {threads}: {step_ms(n, 32):.3f} ms")
        model.set_option("mv_tail_threads", 512)
