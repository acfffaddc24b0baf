"""Pick the lm_head row-scale seed of the PEAKED synthetic weight set (tests/helpers.py::peaked_lm_head) on the CPU oracle:
for each candidate seed run ViT + prefill + 16 greedy steps of ds-7b (bf16 policy) and print the top-2 gaps in bf16 ulps.
The GPU test (tests/test_gpu_parity_batched.py::test_peaked_logits_weight_set_is_token_identical) re-checks the gaps itself.
Usage: python tools/peaked_seed_search.py [model] [n_steps] [beta] [seed ...]   (~27 GB of RAM for ds-7b)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import sampling  # noqa: E402
from oracle.model import DetikzifyOracle  # noqa: E402
from oracle.synth import make_weights  # noqa: E402
from detikzify_amd.model.config import preset  # noqa: E402
from detikzify_amd.model.processing import DetikzifyProcessor  # noqa: E402,F401
from tests.helpers import peaked_lm_head, sketch_image  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "detikzify-ds-7b"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
beta = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
seeds = [int(s) for s in sys.argv[4:]] or [0, 1, 2, 3]
c = preset(name)
c.max_positions = 512
cfg = c.oracle_dict()
t0 = time.perf_counter()
w = make_weights(cfg, 1234)
print(f"weights {time.perf_counter() - t0:.0f} s", flush=True)
from pathlib import Path as _P  # noqa: E402
from detikzify_amd.model import _checkpoint_image_processor, _synthetic_tokenizer  # noqa: E402
tok = _synthetic_tokenizer(c)
proc = DetikzifyProcessor(image_processor=_checkpoint_image_processor(_P(name), c), tokenizer=tok, image_seq_len=c.num_patches,
                          image_token=tok.convert_ids_to_tokens(c.patch_token_id))
enc = proc(images=sketch_image(0, 224), return_tensors="pt")
ids, px = enc.input_ids[0], enc.pixel_values
img_tok, eos = cfg["image_token_id"], 2
o = DetikzifyOracle(cfg, w, precision="bf16")
base_head = w["lm_head.weight"]
h = o.llm.forward(o.input_embeds(ids, px[0]))
snap = (list(o.llm.k), list(o.llm.v), o.llm.pos)
print(f"prefill {time.perf_counter() - t0:.0f} s", flush=True)
for seed in seeds:
    w["lm_head.weight"] = peaked_lm_head(base_head, beta, seed)
    o.llm.k, o.llm.v, o.llm.pos = list(snap[0]), list(snap[1]), snap[2]
    logits = o.llm.logits(h[-1])
    gaps, toks = [], []
    for i in range(n_steps):
        m = sampling.mask_scores(logits, [img_tok], [eos], i == 0)
        top2 = torch.topk(m, 2)[0]
        gaps.append(float(top2[0] - top2[1]) / (float(top2[0].abs()) * 2.0 ** -7))
        t = int(torch.argmax(m))
        toks.append(t)
        logits = o.step(t)
    print(f"beta {beta} seed {seed}: distinct tokens {len(set(toks))}; min gap {min(gaps):.1f} ulps; gaps {' '.join(f'{g:.0f}' for g in gaps)}; tokens {toks}; {time.perf_counter() - t0:.0f} s", flush=True)
