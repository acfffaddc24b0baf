import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def _cpu_share():
    """CPUs this process may actually use: the affinity mask, cut to the cgroup's CFS quota.  The GPU boxes give a container 16 CPUs'
    worth of quota (cpu.max = 1600000 100000) on a 256-thread host, and torch sizes its intra-op pool from the HOST: 128 threads fighting
    for 16 CPUs ran the CPU oracle's fp32 Linear 3.5 x slower than 16 threads do (61.6 vs 17.5 ms for 243 x 4096 x 11008,
    profiles/r06q_threads_probe.txt) — the round-6 GPU suite took 894 s on one box and 1199.6 s on the next, against the driver's 1200 s."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except Exception:
        try:
            quota = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            period = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if quota > 0 and period > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except Exception:
            pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import os
    if not os.environ.get("OMP_NUM_THREADS"):       # the oracle is torch on the CPU: its pool follows what the container may use
        try:
            import torch
            share = _cpu_share()
            if torch.get_num_threads() > share:
                torch.set_num_threads(share)
        except Exception:  # pragma: no cover - torch-less environment
            pass
    # threads + a device: a test that hangs must fail on its own instead of taking the whole run with it
    # (pytest-timeout; the slowest test, the ds-1.3b CPU-oracle comparison, takes ~15 s)
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 600


# Order of the GPU tests (round 4's driver run was cut at 1200 s in the middle of the OLDEST tests' full-depth comparisons and never
# reached that round's own kernels): the newest kernels' op-level tests first, then the other light tests, and the full-size
# CPU-oracle comparisons LAST, grouped by model so tests/fullsize.py builds each model's host side (weights read back, ViT + prefix
# prefill of both oracles) exactly once.  Within a group the collection order is kept.
_FIRST_MODULES = ("test_gpu_parity_attn.py", "test_gpu_parity_mx.py", "test_gpu_parity_mv.py")
_FULL_SIZE = (          # (substring of the node id, group): group = position in the run's tail
    ("test_ds13b_matches_cpu_oracle", 0), ("test_vit_error_grows_block_by_block", 0), ("test_long_context_properties_ds13b", 0),
    ("test_full_size_incremental_equals_batched[detikzify-ds-1.3b]", 0),
    ("test_decoder_error_grows_with_depth", 1), ("test_greedy_margins_are_not_biased", 1),
    ("test_full_size_incremental_equals_batched[detikzify-ds-7b]", 2), ("test_headline_models_match_cpu_oracle[detikzify-ds-7b", 2),
    ("test_few_slot_contexts_match_cpu_oracle[detikzify-ds-7b", 2), ("test_long_context_steps_match_cpu_oracle", 2),
    ("test_peaked_logits_weight_set_is_token_identical[detikzify-ds-7b", 2), ("test_batched_headline_matches_cpu_oracle[detikzify-ds-7b", 2),
    ("test_peaked_logits_weight_set_is_token_identical[detikzify-ds-1.3b", 0), ("test_peaked_logits_weight_set_is_token_identical[detikzify-cl-7b", 3),
    ("test_peaked_logits_weight_set_is_token_identical[detikzify-v2-8b", 4),
    ("test_headline_models_match_cpu_oracle[detikzify-cl-7b", 3), ("test_few_slot_contexts_match_cpu_oracle[detikzify-cl-7b", 3),
    ("test_batched_headline_matches_cpu_oracle[detikzify-cl-7b", 3), ("test_mxfp8_activations_against_bf16_activations", 3),
    ("test_full_size_incremental_equals_batched[detikzify-v2-8b]", 4), ("test_headline_models_match_cpu_oracle[detikzify-v2-8b", 4),
    ("test_v2_8b_batched_matches_cpu_oracle", 4),
)


def _gpu_order_key(item):
    nid = item.nodeid
    for sub, group in _FULL_SIZE:
        if sub in nid:
            return (2, group)
    for k, mod in enumerate(_FIRST_MODULES):
        if mod in nid:
            return (0, k)
    return (1, 0)


def pytest_collection_modifyitems(config, items):
    """a plain `pytest` on a machine without a HIP device skips the gpu-marked tests instead of erroring in dtk_create (the GPU
    box, and the driver's `-m gpu` run there, see a device and run them all; DTK_FORCE_GPU_TESTS=1 runs them regardless)"""
    import os
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if gpu_items:                       # stable sort: CPU tests keep their place in front, GPU tests follow in the order above
        rest = [it for it in items if "gpu" not in it.keywords]
        items[:] = rest + sorted(gpu_items, key=_gpu_order_key)
    if os.environ.get("DTK_FORCE_GPU_TESTS") == "1":
        return
    try:
        import torch
        have_gpu = torch.cuda.is_available() and torch.cuda.device_count() > 0
    except Exception:  # pragma: no cover - torch-less environment
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs a HIP device (MI355X): none visible on this machine")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
