"""Seeded synthetic inputs of the measurement protocol: no real sketches exist offline, so the bench, the
smoke test and the parity tests all draw the same deterministic "sketch"."""
from __future__ import annotations

import numpy as np
from PIL import Image, ImageDraw


def sketch_image(seed: int = 0, size: int = 224) -> Image.Image:
    """white canvas with 12 random black poly-lines, 2 px wide (SURVEY.md §8d synthetic input)"""
    rng = np.random.default_rng(seed)
    img = Image.new("RGB", (size, size), "white")
    d = ImageDraw.Draw(img)
    for _ in range(12):
        pts = [tuple(int(v) for v in rng.integers(8, size - 8, 2)) for _ in range(int(rng.integers(2, 5)))]
        d.line(pts, fill="black", width=2)
    return img
