#!/usr/bin/env python
"""Root-parallel MCTS for ONE image over all GPUs of a node: every rank runs `--trees` independent searches as one
batched decode on its GPU (detikzify_amd.infer.batching), rank 0 gathers the (score, code) records — the only
collective, a few KB over RCCL — and keeps the best ones.  Semantics: independent trees (the divergence from the
sequential search that sharding over GPUs implies); `--nproc-per-node 1 --trees 1` is the reference search.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        examples/mcts_multi_gpu.py --model /ckpt/detikzify-ds-7b --image sketch.png --trees 32 --expansions 4
"""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from detikzify_amd import dist as ddist  # noqa: E402
from detikzify_amd.infer import DetikzifyPipeline, SyntheticTikzDocument  # noqa: E402
from detikzify_amd.model import load  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", required=True)
ap.add_argument("--image", required=True)
ap.add_argument("--synthetic", type=int, default=None)
ap.add_argument("--trees", type=int, default=32, help="independent trees per GPU (<= 64: one batched decode)")
ap.add_argument("--expansions", type=int, default=4, help="rollouts per tree")
ap.add_argument("--no-latex", action="store_true")
ap.add_argument("--tex-workers", type=int, default=16, help="LaTeX worker processes per rank (compile pool); 0 = compile in the tree's own thread")
ap.add_argument("--keep", type=int, default=5)
args = ap.parse_args()

local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())   # (% : several ranks on one GPU when testing)
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    torch.cuda.set_device(local_rank)
    ddist.init_process_group(os.environ.get("DTK_DIST_BACKEND"))      # default nccl (= RCCL); "gloo" for a control-flow test
model, processor = load(args.model, synthetic=args.synthetic, device_map=local_rank, batch_slots=min(64, args.trees) + 1)
pool = None
if args.no_latex:
    kw = dict(document_class=SyntheticTikzDocument)
elif args.tex_workers > 0:      # latexmk / crop / rasterise of finished rollouts in worker processes while the other trees decode
    from detikzify_amd.infer import CompilePool, pooled_document_class
    pool = CompilePool(workers=args.tex_workers)
    pool.warm()
    kw = dict(document_class=pooled_document_class(pool))
else:
    kw = {}
pipe = DetikzifyPipeline(model, processor, **kw)
try:
    best = ddist.root_parallel_search(pipe, args.image, trees=args.trees, expansions_per_tree=args.expansions)
finally:
    if pool is not None:
        pool.close()
if ddist.rank() == 0:
    for score, code in best[-args.keep:][::-1]:      # merge_rollouts sorts ascending (eval.py:106)
        print(f"% score {score:.4f}\n{code}\n")
