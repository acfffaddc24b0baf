#!/usr/bin/env python
"""In-situ microbenchmark of the decode GEMV kernel variants (dtk_bench_gemv): every variant of
every role over all layers of a synthetic model, HIP-event timed.  Prints a table and writes JSON.
    python tools/tune_gemv.py --model detikzify-ds-7b --out gpurun_out/tune_gemv.json"""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from detikzify_amd.model import load  # noqa: E402

ROLES = {0: ("qkv", 12), 1: ("o_proj", 13), 2: ("gate_up", 12), 3: ("down", 13), 4: ("lm_head", 5)}

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="detikzify-ds-7b")
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--out", default="gpurun_out/tune_gemv.json")
args = ap.parse_args()
model, _ = load(args.model, synthetic=1234)
c = model.config
bytes_of = {0: 3 * c.hidden * c.hidden * 2, 1: c.hidden * c.hidden * 2, 2: 2 * c.ffn * c.hidden * 2,
            3: c.hidden * c.ffn * 2, 4: c.vocab * c.hidden * 2}
res = {}
for role, (name, nvar) in ROLES.items():
    for v in range(nvar):
        us = C.c_float()
        best = 1e9
        for _ in range(2):
            model._check(model.lib.dtk_bench_gemv(model._ctx, role, v, args.reps, C.byref(us)), "dtk_bench_gemv")
            best = min(best, us.value)
        gbs = bytes_of[role] / (best * 1e-6) / 1e9
        res[f"{name}/v{v}"] = {"us": best, "GBps": gbs}
        print(f"{name:8s} v{v}: {best:8.2f} us  {gbs:8.1f} GB/s", flush=True)
# Infinity Cache probe: the same layer's weights re-read back to back (fits the 256 MB MALL for
# every role but lm_head) vs the rotating-layer numbers above
for role, (name, _) in ROLES.items():
    if role == 4:
        continue
    us = C.c_float()
    model._check(model.lib.dtk_bench_gemv(model._ctx, role, 0x100, args.reps, C.byref(us)), "dtk_bench_gemv")
    res[f"{name}/same_layer"] = {"us": us.value, "GBps": bytes_of[role] / (us.value * 1e-6) / 1e9}
    print(f"{name:8s} same-layer: {us.value:8.2f} us  {res[name + '/same_layer']['GBps']:8.1f} GB/s", flush=True)
Path(args.out).parent.mkdir(parents=True, exist_ok=True)
Path(args.out).write_text(json.dumps(res, indent=1))
