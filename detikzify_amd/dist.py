"""
Multi-GPU inference (row e): one process per GPU, a full model replica per rank, no traffic during
generation, ONE exchange of finished TikZ strings at the end — the pattern of reference
examples/eval.py:80-83,108-137 (`chunk` striping, `dist.all_gather_object`, `interleave`).

  * shard by image (exact reference semantics): rank r takes items[r::world]
  * one image, N rollouts (root parallelisation, a documented semantic divergence from the
    sequential tree search): rank r grows its own tree for its share of the
    expansions with seed base+rank; all (score, code) pairs are gathered and merged like
    eval.py:106 (sorted by score).  world_size 1 is the unmodified sequential search.

The exchange sends every rank's records to rank 0 (`dist.gather` of a padded uint8 slab after an 8-byte
all_gather of the lengths): over RCCL / xGMI with GPU tensors when the process group's backend is nccl, over gloo
with CPU tensors otherwise (CPU tests).  A few KB per rank, latency bound — there is no all-reduce anywhere on this
path.  The reference's `all_gather_object` (eval.py:132) hands the list to every rank although only rank 0 uses it
(:134-136); `all_ranks=True` gives that behaviour (one all_gather instead of the gather).
"""
from __future__ import annotations

import json
import os
from itertools import count
from typing import Any, Dict, List, Optional, Sequence

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


def gather_objects(obj: Any, dst: int = 0, all_ranks: bool = False) -> Optional[List[Any]]:
    """Every rank contributes one JSON-serialisable object; rank `dst` gets the list in rank order, the others None
    (`all_ranks=True`: every rank gets it).  Two fixed-shape collectives: the lengths (all_gather of one int64 — every
    rank needs the maximum to pad its slab), then the padded uint8 slabs (gather to `dst`, or all_gather)."""
    if world() == 1:
        return [obj]
    dev = _device_for_collectives()
    payload = torch.frombuffer(bytearray(json.dumps(obj).encode()), dtype=torch.uint8).to(dev)
    n = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world())]
    dist.all_gather(sizes, n)
    lengths = [int(s.item()) for s in sizes]
    slab = torch.zeros(max(lengths), dtype=torch.uint8, device=dev)
    slab[: payload.numel()] = payload
    if all_ranks:
        slabs = [torch.zeros_like(slab) for _ in range(world())]
        dist.all_gather(slabs, slab)
    else:
        slabs = [torch.zeros_like(slab) for _ in range(world())] if rank() == dst else None
        dist.gather(slab, gather_list=slabs, dst=dst)
        if slabs is None:
            return None
    return [json.loads(bytes(s[:k].cpu().tolist()).decode()) for s, k in zip(slabs, lengths)]


def tree_seed(seed_base: int, tree: int) -> int:
    """Seed stream of tree `tree` of this rank: seed_base + rank + world * tree.  With one tree per rank that is the
    protocol of SURVEY.md §8d/e to the letter — rank r searches with seed 1000 + r, world size 1 with seed 1000 — and
    further trees of a rank interleave behind it without ever colliding with another rank's."""
    return seed_base + rank() + world() * tree


def placement() -> Dict[str, Any]:
    """Which process group this rank is in and which GPU it drives — bench.py reports it for every rank so that a
    multi-GPU number can be checked for one-rank-per-GPU placement."""
    info: Dict[str, Any] = {"rank": rank(), "world": world(), "backend": dist.get_backend() if world() > 1 else None,
                            "local_rank": int(os.environ.get("LOCAL_RANK", "0"))}
    if torch.cuda.is_available():
        i = torch.cuda.current_device()
        props = torch.cuda.get_device_properties(i)
        info.update(cuda_device=i, device_name=props.name, device_uuid=str(getattr(props, "uuid", "")),
                    pci_bus_id=getattr(props, "pci_bus_id", None))
    info["cpus_allowed"] = len(os.sched_getaffinity(0))        # after pin_to_gpu_numa_node(): the GPU's NUMA node
    return info


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/nodeN/cpulist)"""
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_cpus(pci_bdf: str, sysfs: str = "/sys") -> Optional[List[int]]:
    """CPUs of the NUMA node the GPU at PCI address `pci_bdf` ("0000:c1:00.0") hangs off, None when the platform does not say
    (no such device, numa_node -1 on single-node machines)"""
    try:
        node = int(open(f"{sysfs}/bus/pci/devices/{pci_bdf}/numa_node").read())
        if node < 0:
            return None
        return parse_cpulist(open(f"{sysfs}/devices/system/node/node{node}/cpulist").read()) or None
    except (OSError, ValueError):
        return None


def _device_bdf(i: int) -> Optional[str]:
    try:
        props = torch.cuda.get_device_properties(i)
        return f"{int(getattr(props, 'pci_domain_id', 0)):04x}:{int(props.pci_bus_id):02x}:{int(getattr(props, 'pci_device_id', 0)):02x}.0"
    except (AttributeError, TypeError, ValueError, RuntimeError, AssertionError):
        return None


def share_of_cpus(cpus: Sequence[int], k: int, m: int) -> List[int]:
    """the k-th of m disjoint shares of a NUMA node's CPU list: every contiguous run of ids is cut into m equal pieces and share k
    takes the k-th piece of each — Linux lists a node as "cores, then their SMT siblings" (0-63,128-191), so a share keeps its
    cores' siblings.  Runs shorter than m are left whole (shared by everybody)."""
    cpus = sorted(cpus)
    runs: List[List[int]] = []
    for c in cpus:
        if runs and c == runs[-1][-1] + 1:
            runs[-1].append(c)
        else:
            runs.append([c])
    out: List[int] = []
    for run in runs:
        if len(run) < m:
            out += run
        else:
            per = len(run) // m
            out += run[k * per:(k + 1) * per] if k < m - 1 else run[k * per:]
    return out


def pin_to_gpu_numa_node(device_index: Optional[int] = None, sysfs: str = "/sys", local_world: Optional[int] = None) -> Optional[List[int]]:
    """One rank per GPU means 8 processes x (64 tree threads + reward work) on one host: keep each rank's threads on the NUMA
    node of ITS GPU (PCIe root + memory local to the pinned staging buffers) instead of letting 8 x 64 threads roam over both
    sockets — and, when the launcher says how many ranks share the host (LOCAL_WORLD_SIZE, torchrun), on this rank's OWN share of
    that node's CPUs: the ranks whose GPUs hang off the same node split its cores between them (`share_of_cpus`).  Measured with the
    shipped Python stack over an emulated GPU on the MI355X box's 2 x 64-core host (profiles/r05_host_emulation_gpu_box_pinned.txt):
    8 ranks x 64 trees reach 21.8-22.9 rollouts/s each when they roam and 29.8-31.8 with a CPU set of their own.
    Restricts this process (and every thread it starts later) to those CPUs, intersected with what the process is allowed to use;
    returns the CPU list, or None when nothing was changed (unknown topology, DTK_NO_PIN=1)."""
    if os.environ.get("DTK_NO_PIN") or not torch.cuda.is_available():
        return None
    try:
        i = torch.cuda.current_device() if device_index is None else device_index
    except (RuntimeError, AssertionError):
        return None
    bdf = _device_bdf(i)
    cpus = gpu_numa_cpus(bdf, sysfs) if bdf else None
    if not cpus:
        return None
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if not allowed:
        return None
    if local_world is None:
        try:
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        except ValueError:
            local_world = 1
    if local_world > 1 and not os.environ.get("DTK_PIN_WHOLE_NODE"):
        # The share is taken by LOCAL RANK among the local ranks whose GPUs sit on MY node (taking it by device index gave
        # several ranks that drive ONE device — N gloo ranks on a single GPU, a custom device_map — the same share).  Which device
        # rank j drives: j itself with one rank per GPU (bench.py, the examples); j modulo the device count when there are more ranks
        # than devices (bench.py's gloo control-flow path).  A LOCAL_RANK that contradicts both leaves the whole node to this rank.
        try:
            me = int(os.environ.get("LOCAL_RANK", str(i)))
        except ValueError:
            me = i
        n_dev = max(1, torch.cuda.device_count())
        dev_of = (lambda j: j) if local_world <= n_dev else (lambda j: j % n_dev)
        if 0 <= me < local_world and dev_of(me) == i:
            mates = [j for j in range(local_world) if dev_of(j) == i or gpu_numa_cpus(_device_bdf(dev_of(j)) or "", sysfs) == cpus]
            if me in mates and len(mates) > 1:
                mine = share_of_cpus(allowed, mates.index(me), len(mates))
                if len(mine) >= 4:          # never squeeze a rank onto a handful of CPUs (its tree threads, reward work, compile workers)
                    allowed = mine
    try:
        os.sched_setaffinity(0, allowed)
    except OSError:             # a container that forbids it: placement is advice, never a reason to fail the run
        return None
    return allowed


def merge_rollouts(per_rank: Sequence[Sequence[Sequence[Any]]]) -> List[List[Any]]:
    """[(score, code)...] of every rank -> one list sorted by score (eval.py:106), duplicates dropped"""
    seen, out = set(), []
    for r in per_rank:
        for score, code in r:
            if (score, code) not in seen:
                seen.add((score, code))
                out.append([score, code])
    return sorted(out, key=lambda sc: (sc[0], sc[1]))       # equal scores in code order, not in arrival order (thread timing)


def root_parallel_search(pipeline, image, trees: int, expansions_per_tree: int, seed_base: int = 1000,
                         all_ranks: bool = False, **gen_kwargs) -> Optional[List[List[Any]]]:
    """BASELINE configs 4/5: one image, every rank grows `trees` independent trees as ONE batched decode on its GPU
    (infer/batching.simulate_parallel; trees == 1 is the sequential search), then the single exchange of the path:
    all (score, code) records to rank 0, merged there like eval.py:106 (other ranks return None unless `all_ranks`).
    Tree t of rank r samples with the seed stream `tree_seed(seed_base, t)` = seed_base + r + world * t."""
    from .infer.batching import simulate_parallel
    mine = [[float(score), doc.code] for score, doc in
            simulate_parallel(pipeline, image, trees=trees, expansions_per_tree=expansions_per_tree,
                              seeds=[tree_seed(seed_base, t) for t in range(trees)], **gen_kwargs)]
    gathered = gather_objects(mine, all_ranks=all_ranks)
    return None if gathered is None else merge_rollouts(gathered)


def root_parallel_search_images(pipeline, images: Sequence[Any], trees_per_image: int, expansions_per_tree: int,
                                seed_base: int = 1000, all_ranks: bool = False, **gen_kwargs) -> Optional[List[List[List[Any]]]]:
    """BASELINE config 5 (a batch of images, N rollouts each, 8 GPUs): images are striped over the ranks (images[r::world]),
    each rank searches ITS images concurrently — len(mine) * trees_per_image trees in one batched decode, every image
    encoded once — and one gather returns to rank 0, for every image in input order, its (score, code) records sorted by
    score."""
    from .infer.batching import simulate_parallel_images
    mine = chunk(list(range(len(images))), world())[rank()]
    local: List[List[List[Any]]] = [[] for _ in mine]
    if mine:
        n_trees = len(mine) * trees_per_image
        for k, score, doc in simulate_parallel_images(pipeline, [images[i] for i in mine], trees_per_image, expansions_per_tree,
                                                      seeds=[tree_seed(seed_base, t) for t in range(n_trees)], **gen_kwargs):
            local[k].append([float(score), doc.code])
    gathered = gather_objects(local, all_ranks=all_ranks)
    if gathered is None:
        return None
    return [merge_rollouts([records]) for records in interleave_all(gathered, len(images))]


def sharded_sample(pipeline, images: Sequence[Any], all_ranks: bool = False, **gen_kwargs) -> Optional[List[str]]:
    """Shard by image (exact reference semantics, examples/eval.py:80-83,125): rank r samples images[r::world]; rank 0
    gets the TikZ programs of all images in input order (the others None unless `all_ranks`)."""
    mine = [pipeline.sample(image=img, **gen_kwargs).code for img in chunk(images, world())[rank()]]
    gathered = gather_objects(mine, all_ranks=all_ranks)
    return None if gathered is None else interleave_all(gathered, len(images))


def interleave_all(chunks: Sequence[Sequence[Any]], total: int) -> List[Any]:
    """inverse of chunk() for ragged chunks (len(items) not a multiple of the world size)"""
    out: List[Any] = [None] * total
    for r, c in enumerate(chunks):
        out[r::len(chunks)] = c
    return out
