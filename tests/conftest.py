import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dmft-lanc-ed_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a stuck test must fail by itself long before an outer watchdog gives up on the whole run (pytest-timeout, if present)
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 400


@pytest.fixture(scope="session")
def edgpu_mod():
    import edgpu
    return edgpu
