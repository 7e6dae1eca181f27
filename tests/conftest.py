import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: test needs a real MI355X (run with -m gpu)")


@pytest.fixture
def bit_repeatable(monkeypatch):
    """Engines built inside the test sum the decoder gradient over the gene
    strips in a fixed order (``set_dd_atomics(False)``, `scvae train
    --deterministic`): for tests whose point is that two paths or two runs
    carry IDENTICAL bits.  The plan's default -- fp32 atomics, whose order of
    additions differs from run to run -- is what every oracle test runs."""
    from scvae_amd.engine import Engine
    init = Engine.__init__

    def patched(self, *args, **kwargs):
        init(self, *args, **kwargs)
        self.set_dd_atomics(False)
    monkeypatch.setattr(Engine, "__init__", patched)


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    return torch.device("cuda:0")
