"""util/image_prep.py: the reward path's Pillow work (trim, LANCZOS pad, the processor's BICUBIC resize) in worker processes —
the pixels the vision tower sees must be the ones the in-thread path produces, bit for bit (the in-thread path is pinned against
the reference's own files: tests/golden/image_prep.json, image_processor_v1.json), a dead pool must never cost a reward, and a
parallel search must give the same scores with and without it."""
import os
import signal

import numpy as np
import pytest
import torch
from PIL import Image

from detikzify_amd.util import expand, load
from detikzify_amd.util.image_prep import PrepPool

from .helpers import fake_processor, sketch_image


@pytest.fixture(scope="module")
def pool():
    p = PrepPool(2)
    assert p.warm() >= 1
    yield p
    p.close()


def _figures():
    rng = np.random.default_rng(3)
    noisy = Image.fromarray((rng.random((300, 420, 3)) * 255).astype(np.uint8))
    framed = Image.new("RGB", (420, 420), "white")
    framed.paste(noisy.crop((0, 0, 200, 90)), (37, 120))                   # something to trim: a 200 x 90 patch on white
    rgba = Image.new("RGBA", (128, 96), (255, 255, 255, 0))
    rgba.paste((10, 20, 30, 255), (20, 10, 90, 70))
    return [sketch_image(4, 224), noisy, framed, load(rgba), Image.new("RGB", (64, 64), "white")]


@pytest.mark.parametrize("size", [384, 420, 90])
def test_pooled_pixels_are_the_in_thread_pixels(pool, size):
    ip = fake_processor(512, 12, size).image_processor
    for fig in _figures():
        fig = load(fig)
        arr = pool.expand_and_resize(fig, *ip.resize_spec())
        assert arr is not None and arr.dtype == np.uint8 and arr.shape == (size, size, 3)
        pooled = ip.resized_pixel_values(arr, return_tensors="pt")["pixel_values"]
        inline = ip(images=expand(fig, max(fig.size), do_trim=True), return_tensors="pt")["pixel_values"]
        assert pooled.dtype == inline.dtype and torch.equal(pooled, inline)
    assert pool.jobs >= 5


def test_imagesim_uses_the_pool_and_survives_its_death():
    """ImageSim with a pool attached feeds the tower the same pixels; when the workers die the reward is computed in-thread"""
    from detikzify_amd.evaluate.imagesim import ImageSim

    seen = []

    class Tower:                       # stands where model.model.vision_model does: records what it is given
        def pooled_only(self, pixel_values):
            seen.append(pixel_values.clone())
            return pixel_values.mean(dim=(1, 2, 3))[:, None].repeat(1, 4)

    ip = fake_processor(512, 12, 96).image_processor
    sim = ImageSim(model=Tower(), processor=ip, mode="cos")
    fig, ref = _figures()[2], sketch_image(1, 128)
    base = sim.get_similarity(fig, ref)
    inline_pixels = [t.clone() for t in seen]
    seen.clear()
    p = PrepPool(1)
    try:
        assert p.warm() == 1
        sim.prep_pool = p
        assert sim.get_similarity(fig, ref) == base and p.jobs == 2
        assert all(torch.equal(a, b) for a, b in zip(seen, inline_pixels))
        os.kill(p._all[0][0].pid, signal.SIGKILL)                          # the worker dies
        seen.clear()
        assert sim.get_similarity(fig, ref) == base                       # ... and the reward is still the reward
        assert p.broken and all(torch.equal(a, b) for a, b in zip(seen, inline_pixels))
    finally:
        p.close()


def test_parallel_search_scores_do_not_depend_on_the_pool(monkeypatch):
    """simulate_parallel attaches the process-wide pool to the pipeline's metric (DTK_REWARD_PREP_WORKERS); the search's scores and
    documents are the same with it and without it (scripted device, real generate loop / engine / MCTS / SelfSim glue)"""
    from detikzify_amd.infer import DetikzifyPipeline, SyntheticTikzDocument
    from detikzify_amd.infer.batching import simulate_parallel
    from detikzify_amd.util import image_prep

    from .test_generate_loop import NIMG, VOCAB, ScriptedDevice

    def run(workers):
        monkeypatch.setenv("DTK_REWARD_PREP_WORKERS", str(workers))
        monkeypatch.setattr(image_prep, "_SHARED", None)
        pipe = DetikzifyPipeline(ScriptedDevice(slots=5), fake_processor(VOCAB, NIMG, 64), metric="model", max_length=NIMG + 40,
                                 document_class=SyntheticTikzDocument)
        out = sorted((round(s, 12), d.code) for s, d in simulate_parallel(pipe, sketch_image(9, 96), trees=4, expansions_per_tree=2))
        pool = image_prep._SHARED
        jobs = pool.jobs if pool else 0
        if pool:
            pool.close()
        monkeypatch.setattr(image_prep, "_SHARED", None)
        return out, jobs

    with_pool, jobs = run(2)
    without, none = run(0)
    assert jobs >= 6 and none == 0          # the rollouts' figures went through the workers (identical documents are scored once)
    assert with_pool == without and len(with_pool) == 8
