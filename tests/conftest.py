import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The GPU box has 256 host cpus: the CPU-side reference runs of the parity tests (a 256x320 model)
    # crawl when every small ATen / oneDNN op fans out over all of them (174 s for one step on a busy
    # host against a few seconds with 16 threads).
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
