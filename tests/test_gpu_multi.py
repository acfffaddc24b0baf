"""Row e on real hardware: TWO ranks, one GPU each, torch.distributed nccl (= RCCL over xGMI), through the library entry
points of detikzify_amd.dist.  Skipped on a 1-GPU box (the driver's per-round box has one); it exists so that the first
multi-GPU lease exercises the N > 1 RCCL path as a test and not for the first time inside the scaling bench.  The same
code runs under gloo with two CPU ranks in tests/test_dist_gloo.py.

Reference pattern: examples/eval.py:80-83 (striping), :110-113 (one replica per rank, device_map=RANK), :125-136 (the one
exchange of strings)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

RANK_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
from detikzify_amd import dist as dd
from detikzify_amd.infer import DetikzifyPipeline, SyntheticTikzDocument
from detikzify_amd.model import load
from tests.helpers import sketch_image

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dd.init_process_group("nccl", timeout_s=300)
assert dist.get_backend() == "nccl" and dd.world() == world == 2 and dd.rank() == rank
where = dd.placement()
assert where["cuda_device"] == local
model, proc = load("detikzify-tiny", synthetic=1234, device_map=local, batch_slots=5)
pipe = DetikzifyPipeline(model, proc, metric="fast", document_class=SyntheticTikzDocument, compile_timeout=None,
                         max_length=12 + 40)
images = [sketch_image(20 + i, 96) for i in range(5)]
# (1) shard by image, greedy: rank 0 must receive exactly what one process produces alone, in input order
codes = dd.sharded_sample(pipe, images, do_sample=False)
# (2) root-parallel search: 3 trees per rank as one batched decode, records to rank 0
best = dd.root_parallel_search(pipe, images[0], trees=3, expansions_per_tree=2)
# (3) config 5's shape: images striped over the ranks, 2 trees per image
per_image = dd.root_parallel_search_images(pipe, images[:3], trees_per_image=2, expansions_per_tree=1)
# (4) the reference's all_gather_object form + a device-tensor collective on the same process group
everyone = dd.gather_objects(where, all_ranks=True)
t = torch.full((4,), float(rank + 1), device="cuda")
dist.all_reduce(t)
torch.cuda.synchronize()
out = dict(rank=rank, where=where, everyone=everyone, allreduce=float(t[0]))
if rank == 0:
    alone = [pipe.sample(image=im, do_sample=False).code for im in images]
    out.update(codes=codes, alone=alone, best=best, per_image=per_image)
else:
    assert codes is None and best is None and per_image is None      # records travel to rank 0 only
open(os.path.join({outdir!r}, f"rank{{rank}}.json"), "w").write(json.dumps(out))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on one node (RCCL over xGMI)")
def test_two_ranks_two_gpus_over_rccl(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT.format(root=str(ROOT), outdir=str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         capture_output=True, text=True, timeout=900, env=env, cwd=str(ROOT))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    r0 = json.loads((tmp_path / "rank0.json").read_text())
    r1 = json.loads((tmp_path / "rank1.json").read_text())
    assert r0["allreduce"] == r1["allreduce"] == 3.0
    assert r0["where"]["cuda_device"] == 0 and r1["where"]["cuda_device"] == 1
    assert r0["where"]["backend"] == "nccl"
    ids = {(w["cuda_device"], w.get("pci_bus_id"), w.get("device_uuid")) for w in r0["everyone"]}
    assert len(ids) == 2, "the two ranks must drive two different GPUs"
    assert r0["codes"] == r0["alone"] and len(r0["codes"]) == 5
    assert 1 <= len(r0["best"]) <= 12 and [b[0] for b in r0["best"]] == sorted(b[0] for b in r0["best"])
    assert len(r0["per_image"]) == 3 and all(1 <= len(r) <= 2 for r in r0["per_image"])


def test_single_gpu_world_one_needs_no_process_group():
    """N = 1 is the unmodified path: the dist entry points work without torch.distributed being initialised"""
    from detikzify_amd import dist as dd
    from detikzify_amd.infer import DetikzifyPipeline, SyntheticTikzDocument
    from detikzify_amd.model import load
    from tests.helpers import sketch_image
    model, proc = load("detikzify-tiny", synthetic=1234, batch_slots=3)
    pipe = DetikzifyPipeline(model, proc, metric="fast", document_class=SyntheticTikzDocument, compile_timeout=None,
                             max_length=12 + 30)
    images = [sketch_image(30 + i, 96) for i in range(2)]
    assert dd.sharded_sample(pipe, images, do_sample=False) == [pipe.sample(image=im, do_sample=False).code for im in images]
    best = dd.root_parallel_search(pipe, images[0], trees=2, expansions_per_tree=1)
    assert 1 <= len(best) <= 2 and dd.tree_seed(1000, 0) == 1000 and dd.placement()["world"] == 1


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on one node (RCCL over xGMI)")
def test_bench_two_gpus_runs_the_drivers_command(tmp_path):
    """the exact command the driver's scaling run uses (`python -m torch.distributed.run ... bench.py --gpus N --steps K
    --warmup W`) at N = 2 with the smallest model and short rollouts: one JSON line from rank 0, whole-job totals, both GPUs
    distinct, the MCTS phases (config 4 = 16 rollouts over the ranks, config 5 = images striped over the ranks) present"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29543", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1",
                          "--warmup", "1", "--model", "detikzify-ds-1.3b", "--config5-model", "detikzify-ds-1.3b", "--new-tokens", "48",
                          "--batch", "16", "--config5-images", "4", "--config5-trees", "2", "--config5-expansions", "1", "--probe-tokens", "4"],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=str(ROOT))
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert len({(r["cuda_device"], r.get("device_uuid")) for r in d["ranks"]}) == 2 and all(r["backend"] == "nccl" for r in d["ranks"])
    m = d["mcts"]
    assert m["config4"]["fixed_length"]["rollouts"] == 16 and m["config4"]["fixed_length"]["trees_per_gpu"] == 8
    assert m["config5"]["fixed_length"]["rollouts"] == 4 * 2 and m["config5"]["fixed_length"]["images_per_gpu"] == 2
    lo, hi = m["config5"]["fixed_length"]["per_rank_rollouts_per_sec_min_max"]
    assert 0 < lo <= hi and m["config5"]["fixed_length"]["gather_seconds"] >= 0
