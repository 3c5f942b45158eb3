import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    # the CPU oracle runs beside the GPU: torch's intra-op pool oversubscribes badly on the 256-thread GPU hosts
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    return torch.device("cuda:0")
