"""
Multi-GPU inference (row e): one process per GPU, a full model replica per rank, no traffic during
generation, ONE exchange of finished TikZ strings at the end — the pattern of reference
examples/eval.py:80-83,108-137 (`chunk` striping, `dist.all_gather_object`, `interleave`).

  * shard by image (exact reference semantics): rank r takes items[r::world]
  * one image, N rollouts (root parallelisation, a documented semantic divergence from the
    sequential tree search): rank r grows its own tree for its share of the
    expansions with seed base+rank; all (score, code) pairs are gathered and merged like
    eval.py:106 (sorted by score).  world_size 1 is the unmodified sequential search.

The gather is an RCCL all_gather of a padded uint8 slab on the GPU when the process group's
backend is nccl (xGMI), a gloo all_gather of CPU tensors otherwise (CPU tests): a few KB per
rank, latency bound — there is no all-reduce anywhere on this path.
"""
from __future__ import annotations

import json
import os
from itertools import count
from typing import Any, List, Sequence

import torch
import torch.distributed as dist


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def init_process_group(backend: str = None, timeout_s: int = 3 * 24 * 3600):
    """torchrun-style bootstrap (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the env)."""
    import datetime
    if dist.is_initialized() or int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=timeout_s))


def chunk(items: Sequence[Any], n: int) -> List[List[Any]]:
    """n striped chunks: chunk i = items[i::n]"""
    return [list(items[i::n]) for i in range(n)]


def interleave(chunks: Sequence[Sequence[Any]]) -> List[Any]:
    """inverse of chunk(): c0[0], c1[0], ..., c0[1], ... until a chunk runs out"""
    out: List[Any] = []
    for idx in count():
        try:
            out.extend(c[idx] for c in chunks)
        except IndexError:
            break
    return out


def shard_expansions(total: int, n: int) -> List[int]:
    """split `total` MCTS expansions over n ranks as evenly as possible"""
    base, extra = divmod(total, n)
    return [base + (1 if r < extra else 0) for r in range(n)]


def _device_for_collectives() -> torch.device:
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_objects(obj: Any) -> List[Any]:
    """All ranks contribute a JSON-serialisable object; every rank gets the list (rank order).
    Two fixed-shape collectives: lengths (int64) then a padded uint8 slab."""
    if world() == 1:
        return [obj]
    dev = _device_for_collectives()
    payload = torch.frombuffer(bytearray(json.dumps(obj).encode()), dtype=torch.uint8).to(dev)
    n = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world())]
    dist.all_gather(sizes, n)
    cap = int(max(int(s.item()) for s in sizes))
    slab = torch.zeros(cap, dtype=torch.uint8, device=dev)
    slab[: payload.numel()] = payload
    slabs = [torch.zeros_like(slab) for _ in range(world())]
    dist.all_gather(slabs, slab)
    return [json.loads(bytes(s[: int(k.item())].cpu().tolist()).decode()) for s, k in zip(slabs, sizes)]


def merge_rollouts(per_rank: Sequence[Sequence[Sequence[Any]]]) -> List[List[Any]]:
    """[(score, code)...] of every rank -> one list sorted by score (eval.py:106), duplicates dropped"""
    seen, out = set(), []
    for r in per_rank:
        for score, code in r:
            if (score, code) not in seen:
                seen.add((score, code))
                out.append([score, code])
    return sorted(out, key=lambda sc: sc[0])


def root_parallel_search(pipeline, image, trees: int, expansions_per_tree: int, seed_base: int = 1000,
                         **gen_kwargs) -> List[List[Any]]:
    """BASELINE configs 4/5: one image, every rank grows `trees` independent trees as ONE batched decode on its GPU
    (infer/batching.simulate_parallel; trees == 1 is the sequential search), then the single exchange of the path:
    all (score, code) records to every rank, merged like eval.py:106.  Tree t of rank r samples with seed stream
    seed_base * (r + 1) + t, so the trees of different ranks differ and a run is reproducible for a fixed world size."""
    from .infer.batching import simulate_parallel
    mine = [[float(score), doc.code] for score, doc in
            simulate_parallel(pipeline, image, trees=trees, expansions_per_tree=expansions_per_tree,
                              seed_base=seed_base * (rank() + 1), **gen_kwargs)]
    return merge_rollouts(gather_objects(mine))


def root_parallel_search_images(pipeline, images: Sequence[Any], trees_per_image: int, expansions_per_tree: int,
                                seed_base: int = 1000, **gen_kwargs) -> List[List[List[Any]]]:
    """BASELINE config 5 (a batch of images, N rollouts each, 8 GPUs): images are striped over the ranks (images[r::world]),
    each rank searches ITS images concurrently — len(mine) * trees_per_image trees in one batched decode, every image
    encoded once — and one gather returns, for every image in input order, its (score, code) records sorted by score."""
    from .infer.batching import simulate_parallel_images
    mine = chunk(list(range(len(images))), world())[rank()]
    local: List[List[List[Any]]] = [[] for _ in mine]
    if mine:
        for k, score, doc in simulate_parallel_images(pipeline, [images[i] for i in mine], trees_per_image, expansions_per_tree,
                                                      seed_base=seed_base * (rank() + 1), **gen_kwargs):
            local[k].append([float(score), doc.code])
    per_image = interleave_all(gather_objects(local), len(images))
    return [merge_rollouts([records]) for records in per_image]


def sharded_sample(pipeline, images: Sequence[Any], **gen_kwargs) -> List[str]:
    """Shard by image (exact reference semantics, examples/eval.py:80-83,125): rank r samples images[r::world]; every
    rank gets the TikZ programs of all images in input order."""
    mine = [pipeline.sample(image=img, **gen_kwargs).code for img in chunk(images, world())[rank()]]
    return interleave_all(gather_objects(mine), len(images))


def interleave_all(chunks: Sequence[Sequence[Any]], total: int) -> List[Any]:
    """inverse of chunk() for ragged chunks (len(items) not a multiple of the world size)"""
    out: List[Any] = [None] * total
    for r, c in enumerate(chunks):
        out[r::len(chunks)] = c
    return out
