"""
Image processor + processor (CPU, rows a·P2 / a·P3 of SURVEY.md §8).

DetikzifyImageProcessor restates reference detikzify/model/v1/processing_detikzify.py:162-253 for
the timm config of the v1 tower (:99-124): resize to 384x384 with PIL BICUBIC (resample=3, :119),
rescale by 1/255 (:245), normalise with mean = std = 0.5 (:116-118), HWC -> CHW (:251).
DetikzifyProcessor restates detikzify/model/processing_detikzify.py:69-115: the prompt is
image_token * image_seq_len + text, tokenised with add_special_tokens=False (:33-39,:102-110).
Pure numpy/PIL (the reference needs py3.11 typing.Unpack and timm, absent here).
"""
from __future__ import annotations

import threading
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch
from PIL import Image


class BatchFeature(dict):
    """The subset of transformers.BatchFeature the inference code touches
    (infer/generate.py:179-183,216-217; evaluate/imagesim.py:101)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, *args, **kwargs):
        device = kwargs.get("device")
        dtype = kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif a is not None:
                device = a
        out = BatchFeature()
        for k, v in self.items():
            if isinstance(v, torch.Tensor):
                if torch.is_floating_point(v):
                    v = v.to(device=device, dtype=dtype) if (device is not None or dtype is not None) else v
                elif device is not None:
                    v = v.to(device=device)
            out[k] = v
        return out


class DetikzifyImageProcessor:
    model_input_names = ["pixel_values"]

    def __init__(self, size: Optional[Dict[str, int]] = None, resample: int = 3, do_resize=True,
                 do_rescale=True, rescale_factor: float = 0.00392156862745098, do_normalize=True,
                 image_mean: Sequence[float] = (0.5, 0.5, 0.5), image_std: Sequence[float] = (0.5, 0.5, 0.5)):
        self.size = size or {"height": 384, "width": 384}
        self.resample, self.do_resize = resample, do_resize
        self.do_rescale, self.rescale_factor = do_rescale, rescale_factor
        self.do_normalize = do_normalize
        self.image_mean, self.image_std = list(image_mean), list(image_std)

    def to_dict(self):
        return {"size": dict(self.size), "resample": self.resample, "do_resize": self.do_resize,
                "do_rescale": self.do_rescale, "rescale_factor": self.rescale_factor,
                "do_normalize": self.do_normalize, "image_mean": self.image_mean,
                "image_std": self.image_std, "image_processor_type": "TimmImageProcessor"}

    def _lut(self, channel: int) -> np.ndarray:
        """value of rescale -> normalise for each of the 256 uint8 inputs of one channel (the same numpy expressions)"""
        key = (channel, self.do_rescale, self.rescale_factor, self.do_normalize, self.image_mean[channel], self.image_std[channel])
        cache = self.__dict__.setdefault("_lut_cache", {})
        if key not in cache:
            x = np.arange(256, dtype=np.uint8)
            if self.do_rescale:
                x = (x.astype(np.float64) * self.rescale_factor).astype(np.float32)
            if self.do_normalize:
                mean = np.array(self.image_mean[channel], dtype=x.dtype)
                std = np.array(self.image_std[channel], dtype=x.dtype)
                x = (x - mean) / std
            cache[key] = x
        return cache[key]

    @staticmethod
    def _to_numpy(image) -> np.ndarray:
        if isinstance(image, Image.Image):
            return np.array(image.convert("RGB") if image.mode != "RGB" else image)
        if isinstance(image, torch.Tensor):
            image = image.cpu().numpy()
        arr = np.asarray(image)
        if arr.ndim == 2:
            arr = np.stack([arr] * 3, axis=-1)
        return arr

    def _resize(self, arr: np.ndarray) -> np.ndarray:
        h, w = self.size["height"], self.size["width"]
        if arr.dtype != np.uint8:  # HF to_pil_image rescales float images in [0,1]
            arr = (arr * 255).astype(np.uint8) if arr.max() <= 1.0 else arr.astype(np.uint8)
        pil = Image.fromarray(arr)
        return np.array(pil.resize((w, h), resample=Image.Resampling(self.resample), reducing_gap=None))

    def preprocess(self, images, return_tensors: Optional[str] = None, **_) -> BatchFeature:
        if not isinstance(images, (list, tuple)):
            images = [images]
        out = []
        for im in images:
            arr = self._to_numpy(im)
            if self.do_resize:
                arr = self._resize(arr)
            if arr.dtype == np.uint8 and arr.ndim == 3 and arr.shape[2] == len(self.image_mean):
                # uint8 input (every path through _resize): the per-pixel arithmetic below has only 256 possible inputs per
                # channel, so it is evaluated once per value and gathered — bit-identical, ~15 ms less GIL time per image
                # (the SelfSim reward calls this once per rollout from 32 threads)
                x = np.stack([self._lut(c)[arr[:, :, c]] for c in range(arr.shape[2])])
                out.append(np.ascontiguousarray(x))
                continue
            x = arr
            if self.do_rescale:
                x = (x.astype(np.float64) * self.rescale_factor).astype(np.float32)
            if self.do_normalize:
                mean = np.array(self.image_mean, dtype=x.dtype)
                std = np.array(self.image_std, dtype=x.dtype)
                x = (x - mean) / std
            out.append(np.ascontiguousarray(x.transpose(2, 0, 1)))
        if return_tensors == "pt":
            return BatchFeature(pixel_values=torch.from_numpy(np.stack(out)))
        if return_tensors == "np":
            return BatchFeature(pixel_values=np.stack(out))
        return BatchFeature(pixel_values=out)

    __call__ = preprocess


class DetikzifyProcessor:
    attributes = ["image_processor", "tokenizer"]

    def __init__(self, image_processor, tokenizer=None, image_seq_len: int = 300,
                 image_token: str = "<|reserved_special_token_2|>", model_expects_text: bool = False):
        if image_processor is None:
            raise ValueError("You need to specify an `image_processor`.")
        if tokenizer is None:
            raise ValueError("You need to specify a `tokenizer`.")
        if image_token not in tokenizer.vocab:
            raise ValueError(f"{image_token} needs to be added to the `tokenizer` vocabulary.")
        self.image_processor, self.tokenizer = image_processor, tokenizer
        self.image_token, self.image_seq_len = image_token, image_seq_len
        self.model_expects_text = model_expects_text
        # HF fast tokenizers are not re-entrant (`truncation=True` re-configures the Rust object: "Already borrowed" when
        # another thread encodes or decodes meanwhile); the trees of simulate_parallel share one processor
        self._tok_lock = threading.RLock()

    def __call__(self, text=None, images=None, image_seq_len: Optional[int] = None,
                 add_bos_token: Optional[bool] = None, add_eos_token: Optional[bool] = None,
                 return_tensors: Optional[str] = None, text_kwargs: Optional[Dict[str, Any]] = None,
                 **_) -> BatchFeature:
        if images is None:
            raise ValueError("`images` are expected as arguments to a `DetikzifyProcessor` instance.")
        if isinstance(images, list) and all(isinstance(i, list) and len(i) == 1 for i in images):
            images = [i[0] for i in images]
        if not isinstance(images, (list, tuple)):
            images = [images]
        if text is None:
            text = len(images) * [""]
        elif isinstance(text, str):
            text = [text]
        if len(images) != len(text):
            raise ValueError(f"Received {len(images)} images for {len(text)} prompts. "
                             "Each prompt should be associated with an image.")
        prompts: List[str] = []
        for prompt in text:
            assert self.image_token not in prompt, "Image tokens are added by the processor!"
            if add_bos_token:
                prompt += self.tokenizer.bos_token
            if add_eos_token:
                prompt += self.tokenizer.eos_token
            n = image_seq_len if image_seq_len is not None else self.image_seq_len
            prompts.append(self.image_token * n + prompt)
        tk = {"add_special_tokens": False, "padding": False}
        tk.update(text_kwargs or {})
        tk.pop("padding_side", None)
        image_inputs = self.image_processor(images=images, return_tensors=return_tensors)
        with self._tok_lock:
            enc = self.tokenizer(text=prompts, **tk)
        ids, mask = enc["input_ids"], enc["attention_mask"]
        if return_tensors == "pt":
            ids = torch.tensor(ids, dtype=torch.long) if not isinstance(ids, torch.Tensor) else ids
            mask = torch.tensor(mask, dtype=torch.long) if not isinstance(mask, torch.Tensor) else mask
        return BatchFeature({**image_inputs, "input_ids": ids, "attention_mask": mask})

    def batch_decode(self, *a, **k):
        with self._tok_lock:
            return self.tokenizer.batch_decode(*a, **k)

    def decode(self, *a, **k):
        with self._tok_lock:
            return self.tokenizer.decode(*a, **k)

    @property
    def model_input_names(self):
        return ["input_ids", "attention_mask", "pixel_values"]
