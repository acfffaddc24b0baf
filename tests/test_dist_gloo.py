"""world_size-2 gloo test of the multi-GPU path (row e): striped sharding, the string gather and
the root-parallel merge — the same code the nccl/RCCL path runs with GPU tensors.  CPU only."""
import json
import os
import socket
import sys
from pathlib import Path

import torch.multiprocessing as mp

from detikzify_amd import dist as ddist

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, outdir):
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      LOCAL_RANK=str(rank))
    from detikzify_amd import dist as dd
    from detikzify_amd.infer import DetikzifyGenerator, SyntheticTikzDocument
    from tests.helpers import FakeModel, fake_processor, sketch_image
    dd.init_process_group("gloo", timeout_s=120)
    assert dd.world() == world and dd.rank() == rank
    # (1) shard by image: rank r handles items[r::world]
    items = list(range(7))
    mine = dd.chunk(items, world)[rank]
    gathered = dd.gather_objects([f"tikz-{i}" * (i + 1) for i in mine])        # to rank 0 (north_star; eval.py:134-136 uses it there only)
    assert (gathered is None) == (rank != 0)
    if rank == 0:
        assert dd.interleave(gathered)[:6] == [f"tikz-{i}" * (i + 1) for i in range(6)]
    everywhere = dd.gather_objects([f"tikz-{i}" * (i + 1) for i in mine], all_ranks=True)   # the reference's all_gather_object
    assert dd.interleave(everywhere)[:6] == [f"tikz-{i}" * (i + 1) for i in range(6)]
    assert dd.gather_objects({"r": rank}, dst=1) == ([{"r": 0}, {"r": 1}] if rank == 1 else None)
    assert dd.tree_seed(1000, 0) == 1000 + rank and dd.tree_seed(1000, 2) == 1000 + rank + 2 * world   # 1000 + rank
    assert dd.placement()["world"] == world and dd.placement()["backend"] == "gloo"
    # (2) one image, 6 expansions root-parallel: each rank grows its own tree with seed base+rank
    share = dd.shard_expansions(6, world)[rank]
    gen = DetikzifyGenerator(FakeModel(seed=1000 + rank), fake_processor(), sketch_image(1, 64), metric=None,
                             document_class=SyntheticTikzDocument, max_length=80, compile_timeout=None)
    local = [[float(s), d.code] for s, d in gen.simulate(expansions=share)]
    merged = dd.merge_rollouts(dd.gather_objects(local, all_ranks=True))
    # (3) the library entry points of BASELINE configs 4/5 on the real generate loop / batch engine (scripted device):
    #     3 batched trees per rank, one gather; and shard-by-image sampling
    from detikzify_amd.infer import DetikzifyPipeline
    from tests.test_generate_loop import NIMG, VOCAB, ScriptedDevice
    proc = fake_processor(VOCAB, NIMG)
    pipe = DetikzifyPipeline(ScriptedDevice(slots=4), proc, metric="fast", document_class=SyntheticTikzDocument,
                             max_length=NIMG + 40, compile_timeout=None)
    best = dd.root_parallel_search(pipe, sketch_image(2, 64), trees=3, expansions_per_tree=2, all_ranks=True)
    # fixed seeds (1000 + rank + world * tree) and a tie-break stream per tree: the same merged records on every run
    assert dd.root_parallel_search(pipe, sketch_image(2, 64), trees=3, expansions_per_tree=2, all_ranks=True) == best
    images = [sketch_image(10 + i, 64) for i in range(5)]
    codes = dd.sharded_sample(pipe, images, all_ranks=True, do_sample=False)
    per_image = dd.root_parallel_search_images(pipe, images[:3], trees_per_image=2, expansions_per_tree=2, all_ranks=True)
    # default: the records travel to rank 0 only
    only0 = dd.sharded_sample(pipe, images, do_sample=False)
    assert (only0 == codes) if rank == 0 else (only0 is None)
    only0 = dd.root_parallel_search(pipe, sketch_image(2, 64), trees=2, expansions_per_tree=1)
    assert (only0 is None) == (rank != 0)
    if rank == 0:
        alone = [pipe.sample(image=im, do_sample=False).code for im in images]
        Path(outdir, "merged.json").write_text(json.dumps({
            "n": len(merged), "scores": [m[0] for m in merged], "best": best, "codes": codes, "alone": alone, "per_image": per_image}))
    else:
        Path(outdir, "rank1.json").write_text(json.dumps({"best": best, "codes": codes, "per_image": per_image}))
    import torch.distributed as td
    td.barrier()
    td.destroy_process_group()


def test_two_rank_gather_and_root_parallel_merge(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = json.loads((tmp_path / "merged.json").read_text())
    assert 1 <= res["n"] <= 6 and res["scores"] == sorted(res["scores"])
    other = json.loads((tmp_path / "rank1.json").read_text())
    assert res["best"] == other["best"] and 1 <= len(res["best"]) <= 12          # every rank holds the merged records
    assert [b[0] for b in res["best"]] == sorted(b[0] for b in res["best"])
    assert res["per_image"] == other["per_image"] and len(res["per_image"]) == 3       # 3 images over 2 ranks: 2 + 1
    assert all(1 <= len(r) <= 4 and [x[0] for x in r] == sorted(x[0] for x in r) for r in res["per_image"])
    assert res["codes"] == other["codes"] == res["alone"] and len(res["codes"]) == 5   # input order, ragged shards


def test_chunk_interleave_roundtrip_and_expansion_split():
    items = list(range(10))
    for n in (1, 2, 3, 4, 8):
        ch = ddist.chunk(items, n)
        assert sorted(sum(ch, [])) == items
        assert ddist.interleave(ch) == items      # partial last round is kept, like eval.py:85-93
        assert sum(ddist.shard_expansions(16, n)) == 16
    assert ddist.shard_expansions(16, 8) == [2] * 8 and ddist.shard_expansions(5, 2) == [3, 2]
    assert ddist.gather_objects({"a": 1}) == [{"a": 1}]       # world 1: no process group needed


def test_sharding_matches_the_reference_script():
    """tests/golden/sharding.json: `chunk` / `interleave` cut out of the reference's examples/eval.py:79-93 and run on
    every (#items, world size) up to (9, 4) — ours give the same shards and the same merged order"""
    golden = json.loads((ROOT / "tests" / "golden" / "sharding.json").read_text())
    for key, want in golden.items():
        n, world = (int(v) for v in key.split("/"))
        items = list(range(100, 100 + n))
        chunks = ddist.chunk(items, world)
        assert chunks == want["chunks"], key
        assert ddist.interleave(chunks) == want["interleaved"] == items, key
        assert ddist.interleave_all(chunks, n) == items, key


def test_numa_pinning_helpers(tmp_path):
    """dist.parse_cpulist / gpu_numa_cpus on a fake sysfs tree (the rank-to-NUMA-node pinning of N > 1 runs)"""
    from detikzify_amd import dist as dd
    assert dd.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and dd.parse_cpulist("") == []
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("64-127\n")
    assert dd.gpu_numa_cpus("0000:c1:00.0", str(tmp_path)) == list(range(64, 128))
    (dev / "numa_node").write_text("-1\n")
    assert dd.gpu_numa_cpus("0000:c1:00.0", str(tmp_path)) is None          # single-node machine: nothing to pin to
    assert dd.gpu_numa_cpus("0000:00:00.0", str(tmp_path)) is None          # unknown device
    assert dd.pin_to_gpu_numa_node() is None                                # no GPU here: a no-op


def test_ranks_on_one_numa_node_get_disjoint_cpu_shares(tmp_path, monkeypatch):
    """8 ranks on a 2-socket host (the MI355X box: node0 = 0-63,128-191, node1 = 64-127,192-255, four GPUs per node): every rank is
    pinned to its OWN quarter of its GPU's node — cores and their SMT siblings together, shares disjoint, their union the node —
    and to the whole node when the launcher does not say how many ranks share the host (tools/host_emulation.py --pin measured 8
    ranks x 64 trees at 30-32 rollouts/s each with a CPU set per rank against 22-23 when they roam)"""
    from types import SimpleNamespace

    import torch

    from detikzify_amd import dist as dd
    assert dd.share_of_cpus(list(range(0, 64)) + list(range(128, 192)), 1, 4) == list(range(16, 32)) + list(range(144, 160))
    assert dd.share_of_cpus([0, 1, 2, 3, 4, 5, 6], 2, 3) == [4, 5, 6]               # the last share takes the remainder
    assert dd.share_of_cpus([0, 1, 8, 9, 10, 11], 0, 4) == [0, 1, 8]               # a run shorter than the share count stays whole
    lists = {0: "0-63,128-191\n", 1: "64-127,192-255\n"}
    for n, text in lists.items():
        node = tmp_path / "devices" / "system" / "node" / f"node{n}"
        node.mkdir(parents=True)
        (node / "cpulist").write_text(text)
    for j in range(8):
        dev = tmp_path / "bus" / "pci" / "devices" / f"0000:{0x10 + j:02x}:00.0"
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text(f"{j // 4}\n")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda j: SimpleNamespace(pci_domain_id=0, pci_bus_id=0x10 + j, pci_device_id=0))
    monkeypatch.setattr(dd.os, "sched_getaffinity", lambda pid: set(range(256)))
    pinned = {}
    monkeypatch.setattr(dd.os, "sched_setaffinity", lambda pid, cpus: pinned.__setitem__("cpus", sorted(cpus)))
    monkeypatch.delenv("DTK_NO_PIN", raising=False)
    shares = [dd.pin_to_gpu_numa_node(j, str(tmp_path), local_world=8) for j in range(8)]
    assert all(len(sh) == 32 for sh in shares) and pinned["cpus"] == shares[7]
    assert sorted(c for sh in shares[:4] for c in sh) == dd.parse_cpulist(lists[0]) and sorted(c for sh in shares[4:] for c in sh) == dd.parse_cpulist(lists[1])
    assert shares[1] == list(range(16, 32)) + list(range(144, 160))                 # 16 cores + their 16 siblings
    assert dd.pin_to_gpu_numa_node(5, str(tmp_path), local_world=1) == dd.parse_cpulist(lists[1])       # one rank per host: the whole node
    # four ranks that all drive device 0 of a one-GPU box (bench.py's gloo control-flow path; ADVICE r5: they used to get the SAME share)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    quarters = []
    for r in range(4):
        monkeypatch.setenv("LOCAL_RANK", str(r))
        quarters.append(dd.pin_to_gpu_numa_node(0, str(tmp_path), local_world=4))
    assert all(len(q) == 32 for q in quarters) and sorted(c for q in quarters for c in q) == dd.parse_cpulist(lists[0])
    monkeypatch.setenv("LOCAL_RANK", "3")           # a local rank that does not drive the device it asks about: no split, the whole node
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert dd.pin_to_gpu_numa_node(5, str(tmp_path), local_world=8) == dd.parse_cpulist(lists[1])
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")                                     # two ranks, one GPU per node: each keeps its whole node
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda j: SimpleNamespace(pci_domain_id=0, pci_bus_id=0x10 + 4 * j, pci_device_id=0))
    assert dd.pin_to_gpu_numa_node(1, str(tmp_path)) == dd.parse_cpulist(lists[1])
