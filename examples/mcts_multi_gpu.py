#!/usr/bin/env python
"""Root-parallel MCTS for ONE image over all GPUs of a node: every rank runs `--trees` independent searches as one
batched decode on its GPU (detikzify_amd.infer.batching), rank 0 gathers the (score, code) records — the only
collective, a few KB over RCCL — and keeps the best ones.  Semantics: independent trees (the divergence from the
sequential search that sharding over GPUs implies); `--nproc-per-node 1 --trees 1` is the reference search.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        examples/mcts_multi_gpu.py --model /ckpt/detikzify-ds-7b --image sketch.png --trees 32 --expansions 4

Everything runs under `main()`: the LaTeX compile pool spawns worker processes, and a spawned worker imports this file again
(as `__mp_main__`) — at module level there must be nothing but imports.
"""
import argparse
import importlib
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True)
    ap.add_argument("--image", required=True)
    ap.add_argument("--synthetic", type=int, default=None)
    ap.add_argument("--trees", type=int, default=32, help="independent trees per GPU (<= 64: one batched decode)")
    ap.add_argument("--expansions", type=int, default=4, help="rollouts per tree")
    ap.add_argument("--no-latex", action="store_true")
    ap.add_argument("--tex-workers", type=int, default=16, help="LaTeX worker processes per rank (compile pool); 0 = compile in the tree's own thread")
    ap.add_argument("--metric", default="model", help="reward: 'model' (SelfSim on the device ViT) or 'fast' (compiler diagnostics only)")
    ap.add_argument("--keep", type=int, default=5)
    return ap.parse_args(argv)


def load_model(args, local_rank: int):
    """(model, processor); DTK_EXAMPLE_LOADER="module:function" swaps the loader (the CPU smoke test runs this script on the
    scripted device of the test suite — there is no CPU implementation of the model)"""
    hook = os.environ.get("DTK_EXAMPLE_LOADER")
    if hook:
        module, _, fn = hook.partition(":")
        return getattr(importlib.import_module(module), fn)(args, local_rank)
    from detikzify_amd.model import load
    return load(args.model, synthetic=args.synthetic, device_map=local_rank, batch_slots=min(64, args.trees) + 1)


def main(argv=None) -> int:
    import torch
    from detikzify_amd import dist as ddist
    from detikzify_amd.infer import DetikzifyPipeline, SyntheticTikzDocument
    args = parse_args(argv)
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())   # (% : several ranks on one GPU when testing)
    pool = None
    # the pool FIRST: its workers are spawned from a process that holds no HIP context, no process group and no model yet
    if args.no_latex:
        kw = dict(document_class=SyntheticTikzDocument)
    elif args.tex_workers > 0:      # latexmk / crop / rasterise of finished rollouts in worker processes while the other trees decode
        from detikzify_amd.infer import CompilePool, pooled_document_class
        pool = CompilePool(workers=args.tex_workers)
        pool.warm()
        kw = dict(document_class=pooled_document_class(pool))
    else:
        kw = {}
    try:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            torch.cuda.set_device(local_rank)
            ddist.init_process_group(os.environ.get("DTK_DIST_BACKEND"))      # default nccl (= RCCL); "gloo" for a control-flow test
            ddist.pin_to_gpu_numa_node(local_rank)       # the rank's 64 tree threads + reward work stay on its GPU's NUMA node
        model, processor = load_model(args, local_rank)
        pipe = DetikzifyPipeline(model, processor, metric=args.metric, **kw)
        best = ddist.root_parallel_search(pipe, args.image, trees=args.trees, expansions_per_tree=args.expansions)
    finally:
        if pool is not None:
            pool.close()
    if ddist.rank() == 0:       # the other ranks get None (the gather's destination is rank 0, as examples/eval.py:125 of the reference)
        for score, code in best[-args.keep:][::-1]:      # merge_rollouts sorts ascending (eval.py:106)
            print(f"% score {score:.4f}\n{code}\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
