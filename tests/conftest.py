import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "wind-downscaling-gan_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ops():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from downscaling.engine.hipops import HipOps
    return HipOps("cuda:0")


@pytest.fixture(scope="session")
def ref_ops():
    from oracle.torch_backend import TorchOps
    return TorchOps()
