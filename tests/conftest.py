import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # threads + a device: a test that hangs must fail on its own instead of taking the whole run with it
    # (pytest-timeout; the slowest test, the ds-1.3b CPU-oracle comparison, takes ~15 s)
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 600


def pytest_collection_modifyitems(config, items):
    """a plain `pytest` on a machine without a HIP device skips the gpu-marked tests instead of erroring in dtk_create (the GPU
    box, and the driver's `-m gpu` run there, see a device and run them all; DTK_FORCE_GPU_TESTS=1 runs them regardless)"""
    import os
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
