import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gym-solarpvder-environment_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    # the in-tree library normally travels with the snapshot; (re)build it with the box's hipcc if it
    # is missing or older than its sources -- the same HIP code either way, never a fallback
    from gym_PVDER_amd import _native
    _native.build_native()
    return torch.device("cuda:0")
