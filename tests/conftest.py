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


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
