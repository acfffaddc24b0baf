#!/usr/bin/env python
"""Image -> TikZ on one MI355X (the counterpart of the reference's examples/infer.py).

    python examples/infer.py --model /ckpt/detikzify-ds-7b --image sketch.png [--mcts-seconds 600]
    python examples/infer.py --model detikzify-ds-7b --synthetic 1234 --image sketch.png      # offline: seeded weights
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from detikzify_amd.infer import DetikzifyPipeline, SyntheticTikzDocument  # noqa: E402
from detikzify_amd.model import load  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", required=True)
ap.add_argument("--image", required=True)
ap.add_argument("--synthetic", type=int, default=None, help="seed for synthetic weights (no checkpoint)")
ap.add_argument("--mcts-seconds", type=int, default=0, help="> 0: MCTS for this long instead of one sample")
ap.add_argument("--no-latex", action="store_true", help="score with the synthetic renderer instead of latexmk")
args = ap.parse_args()

model, processor = load(args.model, synthetic=args.synthetic)
kw = dict(document_class=SyntheticTikzDocument) if args.no_latex else {}
pipe = DetikzifyPipeline(model, processor, **kw)
if args.mcts_seconds > 0:
    best = None
    for score, doc in pipe.simulate(args.image, timeout=args.mcts_seconds):
        if best is None or score > best[0]:
            best = (score, doc)
            print(f"# score {score:.4f}", file=sys.stderr)
    print(best[1].code if best else "")
else:
    print(pipe.sample(args.image).code)
