#!/usr/bin/env python
"""
bench.py — TikZ tokens/sec (+ MCTS rollouts/sec) of the MI355X-native DeTikZify hot path.

Metric (BASELINE.json): TikZ tokens/sec + MCTS rollouts/sec, detikzify-ds-7b, 1 image, N MI355X.
One "step" = one rollout of the hot path through the product API (model.generate, the call
DetikzifyGenerator.generate makes, reference infer/generate.py:218-227): image already preprocessed
on the host -> ViT (666 GF) -> projector -> 243-token prefill -> 512 decoded tokens (EOS suppressed:
fixed work) with bad_words/begin-suppress processors, one D2H per token.  `value` is
generated tokens / wall time of the whole step loop (ViT + prefill INCLUDED); the decode-only rate and
the prefill time are reported beside it.  N > 1 (torchrun, one rank per GPU): every rank runs its own
independent rollouts on a full replica (root-parallel rollouts), the generated token
strings are gathered to all ranks over RCCL inside the timed region; scaling is weak.

Extra objects: `roofline` (dominant kernel = the fused RMSNorm + gate/up GEMV + SiLU·mul kernel,
44 % of the weight bytes; duration measured live with HIP events on the library's stream in a probe
pass of plain launches; HBM peak 8 TB/s) and `cpu_baseline` (the CPU oracle timed on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="detikzify-ds-7b")
    ap.add_argument("--new-tokens", type=int, default=512)
    ap.add_argument("--sample", action="store_true", help="sampling decode (T=.8, p=.95) instead of greedy")
    ap.add_argument("--reuse", action="store_true", help="SURVEY §8 f1: reuse image embeds / prefix KV across rollouts")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=8)
    ap.add_argument("--probe-tokens", type=int, default=64)
    ap.add_argument("--weight-format", default="bf16", choices=["bf16", "fp8"],
                    help="fp8 = e4m3 decoder weights with per-row 2^e scales (BASELINE config 5, cl-7b)")
    ap.add_argument("--skip-batched", action="store_true", help="allocate the --batch slots (for --mcts-trees) but skip the "
                    "batched_rollouts phase itself")
    ap.add_argument("--batch-images", type=int, default=1, help="spread the rollouts of the batched phase over this many "
                    "different images (BASELINE config 5: 8 images x 4 rollouts = --batch 32 --batch-images 8)")
    ap.add_argument("--batch", type=int, default=64, help="independent rollouts decoded as one batch per GPU in the "
                    "extra 'batched_rollouts' phase (0 = skip); the headline value stays batch 1")
    ap.add_argument("--mcts-trees", type=int, default=0, help="optional extra phase: root-parallel MCTS (reference search logic per "
                    "tree, SelfSim reward on the device ViT, LaTeX replaced by the synthetic renderer) with this many trees per GPU")
    ap.add_argument("--mcts-expansions", type=int, default=3, help="rollouts per tree in the --mcts-trees phase")
    return ap.parse_args()


def cpu_baseline(model, cfg, ids, n_tokens):
    """The oracle (a port: HF LlamaModel restated, oracle/llama.py) timed on the host cores:
    greedy decode steps at context 243+ on the SAME weights (copied back from the device)."""
    import torch
    from oracle import sampling
    from oracle.llama import LlamaOracle
    t0 = time.perf_counter()
    import psutil
    names = [n for n in model.tensor_names() if n.startswith(("model.layers.", "model.norm", "lm_head", "model.embed"))]
    total = sum(model.lib.dtk_tensor_numel(model._ctx, n.encode()) for n in names)
    fp32 = psutil.virtual_memory().available > 6 * total + (8 << 30)   # fp32 copies when the host has the RAM
    shapes = {"down_proj": (cfg["hidden"], cfg["ffn"]), "gate_proj": (cfg["ffn"], cfg["hidden"]),
              "up_proj": (cfg["ffn"], cfg["hidden"])}
    w = {}
    for n in names:
        t = model.read_tensor(n)
        if n.endswith("layernorm.weight") or n == "model.norm.weight":
            w[n] = t.float()
            continue
        key = n.split(".")[-2]
        shape = shapes.get(key, (cfg["vocab"], cfg["hidden"]) if n in ("model.embed_tokens.weight", "lm_head.weight")
                           else (cfg["hidden"], cfg["hidden"]))
        w[n] = t.view(*shape).float() if fp32 else t.view(*shape)
    w["model.embed_tokens.weight"] = w["model.embed_tokens.weight"].float()
    llm = LlamaOracle(cfg, w, precision="bf16")
    t_load = time.perf_counter() - t0
    with torch.no_grad():
        h = llm.forward(llm.embed(ids))            # 243-token prefix, not timed (text-only: no CPU ViT)
        logits = llm.logits(h[-1])
        t1 = time.perf_counter()
        done = 0
        for i in range(n_tokens):
            tok = sampling.greedy(logits, [cfg["image_token_id"]], [], False)
            logits = llm.logits(llm.forward(llm.embed(torch.tensor([tok])))[-1])
            done += 1
            if time.perf_counter() - t1 > 30.0 and done >= 2:
                break
        dt = time.perf_counter() - t1
    return {
        "value": done / dt, "unit": "tokens/s", "cores": torch.get_num_threads(),
        "host_cpus": len(os.sched_getaffinity(0)), "kind": "port",
        "sample": f"{done} greedy decode steps at context {ids.numel()}+ of the {cfg['layers']}-layer d={cfg['hidden']} decoder "
                  f"({'bf16 weights upcast to fp32, fp32 GEMV' if fp32 else 'bf16 weights, native bf16 GEMV'}; prefix prefill and weight copy-back "
                  f"({t_load:.0f} s) not timed; no CPU ViT)",
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    from detikzify_amd import dist as ddist
    from detikzify_amd.model import load
    from detikzify_amd.util import expand
    from tests.helpers import sketch_image

    backend = os.environ.get("DTK_DIST_BACKEND", "nccl")     # "gloo": control-flow test of N ranks on one GPU
    n_dev = max(1, torch.cuda.device_count())
    if world > 1:
        if local_rank >= n_dev and backend == "nccl":
            raise SystemExit(f"rank {rank}: local_rank {local_rank} but only {n_dev} GPUs")
        local_rank = local_rank % n_dev
        torch.cuda.set_device(local_rank)
        ddist.init_process_group(backend, timeout_s=1800)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    model, proc = load(args.model, synthetic=1234, device_map=local_rank, batch_slots=min(65, min(64, args.batch) + max(1, args.batch_images)) if args.batch > 1 else 0,   # + a prefix-cache slot per image
                       weight_format=args.weight_format)
    model.reuse_prefix = bool(args.reuse)
    cfg = model.config
    img = sketch_image(0, 224)
    img = expand(img, max(img.size), do_trim=True)                 # P1 (host)
    enc = proc(images=img, return_tensors="pt")                    # P2/P3 (host): pixels resident before timing
    ids, px = enc.input_ids, enc.pixel_values
    T0 = ids.shape[1]
    n_new = args.new_tokens
    gen_kw = dict(pixel_values=px, bad_words_ids=[[cfg.image_token_id]], begin_suppress_tokens=[cfg.eos_token_id],
                  suppress_tokens=[cfg.eos_token_id], max_new_tokens=n_new, eos_token_id=-1)
    if args.sample:
        gen_kw.update(do_sample=True, temperature=0.8, top_p=0.95, top_k=0)
    else:
        gen_kw.update(do_sample=False)

    def rollout(i):
        out = model.generate(input_ids=ids, seed=1000 + rank + 7919 * i, **gen_kw)
        assert out.shape[1] == T0 + n_new
        return out[0, T0:]

    def fence():
        model.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        rollout(-1 - i)
    fence()
    per_step, prefill_ms, vit_ms = [], [], []
    t0 = time.perf_counter()
    codes = []
    for i in range(args.steps):
        ts = time.perf_counter()
        toks = rollout(i)
        per_step.append(time.perf_counter() - ts)
        st = model.stats()
        prefill_ms.append(st["last_prefill_ms"]); vit_ms.append(st["last_vit_ms"])
        codes.append(proc.decode(toks, skip_special_tokens=True))
    if world > 1:   # the path's one exchange: finished TikZ strings to every rank (rank 0 scores them)
        gathered = ddist.gather_objects(codes)
        assert len(gathered) == world
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_tokens = world * args.steps * n_new
    value = total_tokens / elapsed
    st = model.stats()
    W, Kb = st["weight_bytes_per_token"], st["kv_bytes_per_ctx_token"]
    mean_ctx = T0 + (n_new - 1) / 2.0
    bytes_per_token = W + Kb * mean_ctx
    dec_s = [s - p / 1e3 for s, p in zip(per_step, prefill_ms)]
    decode_tok_s = n_new / (sum(dec_s) / len(dec_s))
    result = {
        "metric": "tikz_tokens_per_sec", "value": value, "unit": "tokens/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.weight_format == "bf16" else "fp8-e4m3 weights / bf16 activations", "data": "synthetic",
        "config": {"workload": f"{args.model} (synthetic weights, seed 1234), 1 image 224x224->384x384, 243-token prefix, "
                               f"{'sampling T=.8 p=.95' if args.sample else 'greedy'} decode of {n_new} tokens per rollout, "
                               f"batch 1 per GPU, hipGraph per token" + (", image/prefix reuse" if args.reuse else ""),
                   "tokens_per_rollout": n_new, "prefix_tokens": T0, "rollouts_per_gpu": args.steps},
        "rollouts_per_sec": world * args.steps / elapsed,
        "decode_tokens_per_sec_per_gpu": decode_tok_s,
        "prefill_ms": sum(prefill_ms) / len(prefill_ms), "vit_ms": sum(vit_ms) / len(vit_ms),
        "decode_step": {"algorithmic_bytes_per_token": bytes_per_token, "achieved_GBps": bytes_per_token * decode_tok_s / 1e9,
                        "frac_of_hbm_peak": bytes_per_token * decode_tok_s / 1e9 / HBM_PEAK_GBS,
                        "roofline_tokens_per_sec": HBM_PEAK_GBS * 1e9 / bytes_per_token},
    }

    # ---- extra phase: B independent rollouts per GPU decoded as ONE batch (root-parallel trees of one
    # GPU): the weights are streamed once per step for all B sequences
    if args.batch > 1 and not args.skip_batched:
        import threading
        from detikzify_amd.infer.batching import BatchEngine
        engine = BatchEngine(model, max_batch=args.batch)
        try:
            # MCTS rollouts sample with the pipeline's defaults (temperature .8, top-p .95: generate.py:362-364)
            mcts_kw = {**gen_kw, "do_sample": True, "temperature": 0.8, "top_p": 0.95, "top_k": 0}

            n_img = max(1, min(args.batch_images, args.batch))
            px_of = [px] + [proc(images=expand(sketch_image(100 + k, 224), 224, do_trim=True), return_tensors="pt").pixel_values
                            for k in range(1, n_img)]

            def one(i):
                model.generate(input_ids=ids, seed=5000 + rank * 100 + i, **{**mcts_kw, "pixel_values": px_of[i % n_img]})
            for rep_i in range(2):          # first pass warms the batch graph up
                fence()
                tb = time.perf_counter()
                engine.expect(args.batch)      # the B rollouts start together: first step once all have joined
                ths = [threading.Thread(target=one, args=(i,)) for i in range(args.batch)]
                [t.start() for t in ths]
                [t.join() for t in ths]
                fence()
                t_end = time.perf_counter()
                phases = {"start_to_first_step_ms": round(1e3 * (engine.t_first_launch - tb), 1),
                          "last_collect_to_end_ms": round(1e3 * (t_end - engine.t_last_collect), 1)}
                tb = t_end - tb
            if world > 1:
                t = torch.tensor([tb], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                tb = float(t.item())
            mean_ctx_b = T0 + (n_new - 1) / 2.0
            bytes_step = W + args.batch * Kb * mean_ctx_b
            result["batched_rollouts"] = {
                "batch_per_gpu": args.batch, "images_in_flight": n_img, "prefix_encodes_both_passes": engine.prefix_encodes,
                "rollouts_per_sec": world * args.batch / tb,
                "tokens_per_sec": world * args.batch * n_new / tb, "ms_per_batch": 1e3 * tb,
                "decode_steps": engine.steps, "algorithmic_bytes_per_step": bytes_step,
                # one lock-step decode step per generated token; joins / forks / host time are inside tb, so this is the
                # end-to-end fraction of the decode-step HBM roofline (W once per step + every sequence's own KV)
                "achieved_GBps": bytes_step * n_new / tb / 1e9,
                "frac_of_hbm_peak": bytes_step * n_new / tb / 1e9 / HBM_PEAK_GBS,
                "roofline_rollouts_per_sec": world * args.batch * HBM_PEAK_GBS * 1e9 / (bytes_step * n_new),
                "prefix_sharing": bool(engine.share_prefix),
                "engine_seconds": {"wait": round(engine.t_wait, 3), "launch": round(engine.t_launch, 3), "prefill": round(engine.t_prefill, 3),
                                   "host_bound_steps": engine.host_bound_steps, **phases},
                "decode": "sampling T=.8 top_p=.95 (DetikzifyPipeline defaults), 512 tokens, EOS suppressed",
                "note": "B independent rollouts (own KV slot, seed) per GPU through model.generate from B threads; one "
                        f"dtk_decode_batch step serves all of them; the {T0}-token image prefix is encoded once, its KV "
                        "forked into each slot (bit-identical to a full prefill, SURVEY f1) and read from one copy"}
        except Exception as e:
            result["batched_rollouts"] = {"error": repr(e)}
        finally:
            engine.close()

    # ---- optional phase: the MCTS loop itself (detikzify_amd.infer: DetikzifyGenerator per tree, unchanged reference
    # semantics), `trees` independent trees per GPU decoded as one batch, reward = SelfSim on the device ViT of the
    # SYNTHETIC renderer's image (no TeX offline: stub-reward number, never to be mixed with real-LaTeX numbers)
    if args.mcts_trees > 1 and args.batch > 1:
        try:
            from detikzify_amd.infer import DetikzifyPipeline, SyntheticTikzDocument
            from detikzify_amd.infer.batching import simulate_parallel
            from tests.helpers import sketch_image as _sk
            trees = min(args.mcts_trees, args.batch)
            pipe = DetikzifyPipeline(model, proc, metric="model", document_class=SyntheticTikzDocument,
                                     max_length=T0 + min(n_new, 256))
            pipe.metric.cache_reference = True           # f1: the reference image's features are computed once
            img = _sk(0, 224)
            vit_before = model.stats()["vit_images"]
            fence()
            tm = time.perf_counter()
            res = list(simulate_parallel(pipe, img, trees=trees, expansions_per_tree=args.mcts_expansions))
            fence()
            tm = time.perf_counter() - tm
            if world > 1:
                t = torch.tensor([tm], dtype=torch.float64, device=red_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                tm = float(t.item())
            result["mcts_stub_reward"] = {
                "trees_per_gpu": trees, "expansions_per_tree": args.mcts_expansions, "rollouts": world * len(res),
                "rollouts_per_sec": world * len(res) / tm, "seconds": tm, "max_length": T0 + min(n_new, 256),
                "vit_passes": model.stats()["vit_images"] - vit_before, "engine": getattr(model, "last_batch_stats", None),
                "tokens_generated": getattr(model, "last_batch_stats", {}).get("tokens_out"),
                "reward": "SelfSim (device ViT) of SyntheticTikzDocument renderings; LaTeX absent offline",
                "scores_min_max": [float(min(s for s, _ in res)), float(max(s for s, _ in res))] if res else None}
        except Exception as e:
            result["mcts_stub_reward"] = {"error": repr(e)}

    if rank == 0:
        # ---- roofline of the dominant kernel: probe pass (plain launches, HIP events around the kernel)
        roof = {"bound": "hbm", "kernel": "k_gemv<PRO_RMSNORM,EPI_SWIGLU> (post_attention_layernorm + gate/up GEMV + SiLU*mul)",
                "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
        try:
            model.set_graph_mode(2)
            before = model.stats()
            model.generate(input_ids=ids, **{**gen_kw, "max_new_tokens": args.probe_tokens})
            after = model.stats()
            model.set_graph_mode(1)
            n = after["probe_kernel_launches"] - before["probe_kernel_launches"]
            ms = after["probe_kernel_ms_sum"] - before["probe_kernel_ms_sum"]
            if n > 0 and ms > 0:
                # `achieved` uses the RAW interval between the two events (conservative: an event pair with nothing in
                # between already reads empty_event_pair_us on this stream, and rocprofv3's start->end for the same
                # kernel — profiles/r01_kernel_stats.csv, rocprofv3_avg_us below — is ~8 % shorter than the interval)
                pair_ms = float(after.get("probe_event_pair_ms", 0.0) or 0.0)
                avg_ms = ms / n
                ach = after["probe_kernel_bytes"] / (avg_ms * 1e-3) / 1e9
                roof.update(achieved=ach, frac=ach / HBM_PEAK_GBS, avg_launch_us=avg_ms * 1e3, empty_event_pair_us=pair_ms * 1e3,
                            bytes_per_launch=after["probe_kernel_bytes"], launches_timed=n)
        except Exception as e:  # the bench line must still be printed
            roof["error"] = repr(e)
            model.set_graph_mode(1)
        # HBM traffic of that kernel from the separate rocprofv3 --pmc FETCH_SIZE pass committed under
        # profiles/ (x2 gfx950 correction, guides/MI355X_MICROARCH.md §HBM); null if no profile matches
        try:
            import csv
            prof = ROOT / "profiles" / "r01_pmc_fetch.csv"
            if prof.exists() and args.model == "detikzify-ds-7b":
                for row in csv.DictReader(prof.open()):
                    if row["counter"] == "FETCH_SIZE" and row["kernel"].startswith("void k_gemv<1, 3,"):
                        roof["traffic"] = float(row["hbm_read_bytes_per_launch_x2"])
                        roof["traffic_source"] = "profiles/r01_pmc_fetch.csv (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction)"
                        break
                ks = ROOT / "profiles" / "r01_kernel_stats.csv"
                if ks.exists():
                    for row in csv.DictReader(ks.open()):
                        if row["kernel"].startswith("void k_gemv<1, 3,"):
                            roof["rocprofv3_avg_us"] = float(row["avg_us"])     # committed kernel-trace summary of the same command
                            roof["frac_at_rocprofv3_duration"] = roof["bytes_per_launch"] / (roof["rocprofv3_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS if roof.get("bytes_per_launch") else None
                            break
        except Exception:
            pass
        result["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(model, cfg.kernel_dict(), ids[0], args.cpu_tokens)
            except Exception as e:
                result["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": None, "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
