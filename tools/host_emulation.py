#!/usr/bin/env python
"""Is the host fast enough for the GPU?  Runs the shipped Python stack (DetikzifyPipeline -> simulate_parallel -> DetikzifyGenerator
trees -> model.generate -> BatchEngine) on the scripted device of tests/test_generate_loop.py with the GPU emulated by sleeps:
a decode step takes --step-ms, a reward ViT pass 4 ms, joins are free.  Prints the wall time per step next to the emulated step
time; everything above the emulated time is host overhead (GIL hand-offs, reward image work, joins).  No GPU needed.

    python tools/host_emulation.py --trees 32 --step-ms 4.1
    python tools/host_emulation.py --trees 64 --step-ms 5.8
    python tools/host_emulation.py --trees 64 --step-ms 6.2 --procs 8      # 8 ranks on one host (the N = 8 shape of bench.py)

--procs N runs N such processes side by side (one per emulated GPU) and reports every process's rollouts/s and its HOST CPU
seconds per rollout (user + system time of the whole stack): ranks x rollouts/s x CPU-seconds per rollout = the cores an
8-GPU node must spare for the Python side.
"""
import argparse
import sys
import time
import zlib
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import tests.test_generate_loop as T  # noqa: E402
from detikzify_amd.infer import DetikzifyPipeline, SyntheticTikzDocument  # noqa: E402
from detikzify_amd.infer.batching import simulate_parallel  # noqa: E402
from tests.helpers import fake_processor, sketch_image  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--trees", type=int, default=32)
ap.add_argument("--expansions", type=int, default=3)
ap.add_argument("--step-ms", type=float, default=4.1, help="emulated duration of one batched decode step")
ap.add_argument("--new-tokens", type=int, default=256)
ap.add_argument("--procs", type=int, default=1, help="processes side by side, one per emulated GPU")
ap.add_argument("--json", action="store_true", help="one JSON line (used by --procs)")
ap.add_argument("--pin", action="store_true", help="--procs: every process on its own disjoint share of the CPUs (what dist.pin_to_gpu_numa_node does per rank)")
ap.add_argument("--cpus", default="", help="(internal, --pin) CPU list of this process, e.g. 32-63")
args = ap.parse_args()

if args.procs > 1:
    import json
    import os
    import subprocess
    cmd = [sys.executable, __file__, "--trees", str(args.trees), "--expansions", str(args.expansions), "--step-ms", str(args.step_ms),
           "--new-tokens", str(args.new_tokens), "--json"]
    t0 = time.perf_counter()
    all_cpus = sorted(os.sched_getaffinity(0))
    share = max(1, len(all_cpus) // args.procs)
    pins = [["--cpus", ",".join(str(c) for c in all_cpus[i * share:(i + 1) * share])] if args.pin else [] for i in range(args.procs)]
    kids = [subprocess.Popen(cmd + pins[i], stdout=subprocess.PIPE, text=True) for i in range(args.procs)]
    outs = [json.loads([ln for ln in k.communicate()[0].splitlines() if ln.startswith("{")][-1]) for k in kids]
    wall = time.perf_counter() - t0
    rates = [o["rollouts_per_sec"] for o in outs]
    cpu = [o["cpu_seconds_per_rollout"] for o in outs]
    ideal = outs[0]["emulated_gpu_rollouts_per_sec"]
    print(f"{args.procs} processes{' (each pinned to its own ' + str(share) + ' CPUs)' if args.pin else ''} x {args.trees} trees on {len(os.sched_getaffinity(0))} CPUs: per-process rollouts/s min {min(rates):.1f} / max {max(rates):.1f} "
          f"(the emulated GPU alone allows {ideal:.1f}); whole host {sum(rates):.1f} rollouts/s; host CPU per rollout "
          f"{1e3 * sum(cpu) / len(cpu):.1f} ms (user + sys) -> {sum(rates) * sum(cpu) / len(cpu):.1f} cores busy; {wall:.0f} s incl. start-up")
    sys.exit(0)


if args.cpus:
    import os
    os.sched_setaffinity(0, {int(c) for c in args.cpus.split(",")})


def pooled_only(self, pixel_values):
    time.sleep(0.004)       # the device ViT pass of the reward
    return torch.nn.functional.adaptive_avg_pool2d(pixel_values.float(), 4).flatten() + 1.5


def launch(self, active_slots):
    slots = list(active_slots)
    ready = max(time.perf_counter(), getattr(self, "_busy_until", 0.0)) + args.step_ms * 1e-3
    self.bpending.append(({s: self._next(s) for s in slots}, ready))
    self._busy_until = ready


def wait(self):
    step, ready = self.bpending.pop(0)
    delay = ready - time.perf_counter()
    if delay > 0:
        time.sleep(delay)
    return [step.get(s, -1) for s in range(64)]


def next_token(self, s):    # no EOS: every rollout runs to the length budget, a newline token every ~16 tokens
    ctx = self.ctx[s]
    h = zlib.crc32(repr((self.img[s], len(ctx), ctx[-4:], self.samp[s].get("seed", 0))).encode())
    tok = self.newline[(h >> 16) % len(self.newline)] if (h & 0xFFFF) < 0.06 * 65536 else self.plain[(h >> 16) % len(self.plain)]
    ctx.append(tok)
    return tok


T._Vision.pooled_only = pooled_only
T.ScriptedDevice.decode_batch_launch, T.ScriptedDevice.decode_batch_wait, T.ScriptedDevice._next = launch, wait, next_token
proc = fake_processor(T.VOCAB, T.NIMG, 384)
dev = T.ScriptedDevice(slots=args.trees + 1, max_positions=T.NIMG + args.new_tokens + 8)
pipe = DetikzifyPipeline(dev, proc, metric="model", document_class=SyntheticTikzDocument, max_length=T.NIMG + args.new_tokens,
                         compile_timeout=None)
img = sketch_image(3, 224)
import resource  # noqa: E402
for rep in range(2):        # the first pass warms caches (newline table, reference features)
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    res = list(simulate_parallel(pipe, img, trees=args.trees, expansions_per_tree=args.expansions))
    dt = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
st = dev.last_batch_stats
cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
if args.json:
    import json
    print(json.dumps({"rollouts": len(res), "seconds": dt, "rollouts_per_sec": len(res) / dt, "cpu_seconds_per_rollout": cpu_s / max(1, len(res)),
                      "steps": st["steps"], "wall_ms_per_step": 1e3 * dt / st["steps"],
                      "emulated_gpu_rollouts_per_sec": len(res) / (st["steps"] * args.step_ms / 1e3)}))
    sys.exit(0)
print(f"{args.trees} trees x {args.expansions}: {len(res)} rollouts in {dt:.2f} s = {len(res) / dt:.1f} rollouts/s; {st['steps']} steps, "
      f"{1e3 * dt / st['steps']:.2f} ms wall per step for an emulated {args.step_ms} ms step ({st['steps'] * args.step_ms / 1e3:.2f} s of "
      f"'GPU' time); steps that found the 'GPU' idle: {st['host_bound_steps']}; tokens {st['tokens_out']}")
