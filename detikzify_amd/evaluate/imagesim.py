"""
SelfSim — the MCTS reward (row a·W): cosine similarity of the model's own vision-tower pooled
outputs for (rendered image, input image).  Same behaviour as reference
detikzify/evaluate/imagesim.py:21-147 for the v1 models, whose config reports pooling_mode "cos"
(v1/configuration_detikzify.py:11-13): each image -> load -> expand(trim) -> image processor ->
vision_model(pixel_values).pooler_output -> cosine in float64 (:91-125).  `cos_avg` (mean of the
patch features) is kept.  `emd` (the v2 default, :63,:118-123): patch features of both images -> pairwise
cosine distances in float64 -> earth mover's distance with uniform marginals -> 2*tanh(-emd)+1.  The reference
calls POT's `ot.emd2(a=[], b=[], M)` (absent here); with uniform marginals over two equally sized patch sets
the transport polytope's vertices are permutation matrices (Birkhoff), so the exact optimum is the minimum-cost
assignment divided by n — solved with scipy.optimize.linear_sum_assignment.
torchmetrics is absent here: update/compute/reset are restated with plain accumulators (the
reference disables metric state sync on this path anyway, infer/generate.py:373).  The accumulators are
per thread: `simulate_parallel` shares ONE metric between its trees, and each tree runs the reference's
update -> compute -> reset sequence (infer/generate.py:293-298) on its own thread.

The features of the *reference* image are identical for every rollout; `cache_reference=True`
 memoises them by image bytes — output-identical, one ViT pass per rollout saved.
"""
from __future__ import annotations

from typing import Dict, Literal, Union

import math
import threading

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from ..util import expand, load, unwrap_processor


def pairwise_cosine_distance(a: torch.Tensor, b: torch.Tensor) -> np.ndarray:
    """1 - torchmetrics.functional.pairwise_cosine_similarity(a, b): rows are L2-normalised, then a @ b.T"""
    an = a / a.norm(dim=1, keepdim=True)
    bn = b / b.norm(dim=1, keepdim=True)
    return (1.0 - an @ bn.T).cpu().numpy()


def emd2_uniform(M: np.ndarray) -> float:
    """ot.emd2(a=[], b=[], M): optimal-transport cost between two uniform distributions.  Equal sizes: the
    optimum is a permutation (exact); unequal sizes fall back to the LP (scipy linprog / HiGHS)."""
    n, m = M.shape
    if n == m:
        from scipy.optimize import linear_sum_assignment
        r, c = linear_sum_assignment(M)
        return float(M[r, c].sum() / n)
    from scipy.optimize import linprog
    A_eq = np.zeros((n + m, n * m))
    for i in range(n):
        A_eq[i, i * m:(i + 1) * m] = 1.0
    for j in range(m):
        A_eq[n + j, j::m] = 1.0
    b_eq = np.concatenate([np.full(n, 1.0 / n), np.full(m, 1.0 / m)])
    res = linprog(M.reshape(-1), A_eq=A_eq, b_eq=b_eq, bounds=(0, None), method="highs")
    return float(res.fun)


class ImageSim:
    higher_is_better = True

    def __init__(self, model=None, processor=None, mode: Literal["cos", "cos_avg", "emd"] = "cos",
                 preprocess: bool = True, cache_reference: bool = False, **_):
        if mode not in ("cos", "cos_avg", "emd"):
            raise ValueError(f"unknown pooling mode {mode!r}")
        self.model, self.processor = model, processor
        self.mode, self.preprocess = mode, preprocess
        self.cache_reference = cache_reference
        self._ref_cache: Dict[bytes, torch.Tensor] = {}
        self._acc = threading.local()
        self.reset()

    def __str__(self):
        return self.__class__.__name__ + f" ({self.mode.upper().replace('_', '-')})"

    @classmethod
    def from_detikzify(cls, model, processor, mode=None, *args, **kwargs):
        mode = getattr(model.config, "pooling_mode", "emd") if mode is None else mode
        kwargs.pop("sync_on_compute", None)
        return cls(model=model.model.vision_model, processor=unwrap_processor(processor).image_processor,
                   mode=mode, *args, **kwargs)

    # ---- features --------------------------------------------------------------------------------
    def get_vision_features(self, image: Union[Image.Image, str]) -> torch.Tensor:
        image = load(image)
        if self.preprocess:
            image = expand(image, max(image.size), do_trim=True)
        with torch.inference_mode():
            enc = self.processor(images=image, return_tensors="pt")
            if self.mode == "cos" and hasattr(self.model, "pooled_only"):
                return self.model.pooled_only(enc["pixel_values"]).squeeze()    # same value, no patch-feature copy-back
            out = self.model(**enc)
            if self.mode == "cos":
                return out.pooler_output.squeeze()
            if self.mode == "cos_avg":
                return out.last_hidden_state.squeeze().mean(dim=0)
            return out.last_hidden_state.squeeze()

    def _reference_features(self, image) -> torch.Tensor:
        if not self.cache_reference or not isinstance(image, Image.Image):
            return self.get_vision_features(image)
        key = image.tobytes()
        if key not in self._ref_cache:
            self._ref_cache[key] = self.get_vision_features(image)
        return self._ref_cache[key]

    def get_similarity(self, img1=None, img2=None, **_) -> float:
        f1 = self.get_vision_features(img1)
        f2 = self._reference_features(img2)
        if f1.ndim > 1:   # patch features: earth mover's distance over pairwise cosine distances (:118-123)
            return 2.0 * math.tanh(-emd2_uniform(pairwise_cosine_distance(f1.double(), f2.double()))) + 1.0
        return F.cosine_similarity(f1.double(), f2.double(), dim=0).item()

    # ---- metric protocol (update / compute / reset) ---------------------------------------------
    def update(self, img1=None, img2=None, text1=None, text2=None):
        if text1 is not None or text2 is not None:
            pass  # text conditioning needs the TikZero adapter (out of scope)
        a = img1 if isinstance(img1, list) else [img1]
        b = img2 if isinstance(img2, list) else [img2]
        assert len(a) == len(b)
        for x, y in zip(a, b):
            self.score += self.get_similarity(x, y)
            self.n_samples += 1

    def compute(self) -> float:
        return self.score / self.n_samples

    def reset(self):
        self.score, self.n_samples = 0.0, 0

    # metric state of the calling thread (a thread that never called reset() starts from zero)
    @property
    def score(self) -> float:
        return getattr(self._acc, "score", 0.0)

    @score.setter
    def score(self, v: float):
        self._acc.score = v

    @property
    def n_samples(self) -> int:
        return getattr(self._acc, "n_samples", 0)

    @n_samples.setter
    def n_samples(self, v: int):
        self._acc.n_samples = v
