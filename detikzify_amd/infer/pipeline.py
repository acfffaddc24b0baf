"""
DetikzifyPipeline — the sample() / simulate() front-end over DetikzifyGenerator (behaviour of reference
detikzify/infer/generate.py:356-467: sampling defaults temperature 0.8 / top-p 0.95, `metric` "model" = SelfSim,
"fast" = compiler diagnostics, image loading + trimming).
"""
from __future__ import annotations

from typing import Any, Dict, Generator, Literal, Optional, Tuple, Union

from PIL import Image

from ..evaluate.imagesim import ImageSim
from ..util import expand, load
from ..util import unwrap_processor as unwrap
from .generate import DetikzifyGenerator
from .tikz import TikzDocument
from .tree import Numeric


class DetikzifyPipeline:
    def __init__(self, model, processor, temperature: float = 0.8, top_p: float = 0.95, top_k: int = 0,
                 compile_timeout: Optional[int] = 60,
                 metric: Union[Literal["model", "fast"], Any] = "model", **gen_kwargs):
        self.model, self.processor = model, processor
        if metric == "model":      # SelfSim
            # the features of the input image are the same for every rollout: memoised by image bytes (the
            # reference recomputes them per reward, detikzify/evaluate/imagesim.py:91-125 — same value, one ViT pass less)
            self.metric = ImageSim.from_detikzify(model, processor, sync_on_compute=False, cache_reference=True)
        elif metric == "fast":     # compiler diagnostics
            self.metric = None
        else:
            self.metric = metric
        self.gen_kwargs: Dict[str, Any] = {**dict(
            temperature=temperature, top_p=top_p, top_k=top_k,
            max_length=unwrap(processor).tokenizer.model_max_length,
            do_sample=True, compile_timeout=compile_timeout), **gen_kwargs}

    def load(self, image: Union[Image.Image, str], preprocess: bool = True) -> Image.Image:
        image = load(image)
        return expand(image, max(image.size), do_trim=True) if preprocess else image

    def check_inputs(self, image, text):
        assert text is None or hasattr(self.model, "adapter"), "You need to load an adapter for textual inputs!"
        assert image or text, "Either image or text (or both) required!"

    def _generator(self, image, text, preprocess, **kw) -> DetikzifyGenerator:
        self.check_inputs(image, text)
        return DetikzifyGenerator(
            model=self.model, processor=self.processor,
            image=self.load(image, preprocess=preprocess) if image is not None else None,
            text=text, **{**self.gen_kwargs, **kw})

    def sample(self, image=None, text: Optional[str] = None, preprocess: bool = True, **gen_kwargs) -> TikzDocument:
        """One sampled TikZ program for the image."""
        return self._generator(image, text, preprocess, **gen_kwargs).sample()

    def simulate(self, image=None, text: Optional[str] = None, preprocess: bool = True,
                 expansions: Optional[Numeric] = None, timeout: Optional[int] = None, trees: int = 1,
                 **gen_kwargs) -> Generator[Tuple[Numeric, TikzDocument], None, None]:
        """MCTS: yields (score, document) for every rollout until `expansions` / `timeout`.

        `trees` > 1 (not in the reference): that many independent searches of the same image decoded as ONE batch on
        this GPU (infer/batching.py; the model must have been loaded with batch_slots > trees), each with its own
        `expansions` / `timeout` budget — root parallelisation; results arrive in completion order."""
        if trees > 1:
            from .batching import simulate_parallel
            assert preprocess and text is None, "parallel trees take an image and the default preprocessing"
            yield from simulate_parallel(self, image, trees=trees, expansions_per_tree=expansions or None,
                                         mcts_timeout=timeout or None, **gen_kwargs)
            return
        generator = self._generator(image, text, preprocess, metric=self.metric,
                                    mcts_timeout=timeout or None, **gen_kwargs)
        yield from generator.simulate(expansions or None)

    def __call__(self, *args, **kwargs) -> TikzDocument:
        return self.sample(*args, **kwargs)
