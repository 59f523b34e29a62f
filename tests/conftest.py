import os
import sys

# tests/test_gpu_tp.py runs up to 4 tensor-parallel "ranks" as streams of ONE process whose kernels wait for each other:
# every such stream needs its own hardware queue (the HIP runtime multiplexes streams onto 4 by default; 32 = one per stream of torch's pool).  Read at HIP
# initialisation, so it has to be set before the first device call of the test session.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")
    config.addinivalue_line("markers", "slow: minutes of CPU oracle beside the GPU run; only with BD_RUN_SLOW=1")


def pytest_collection_modifyitems(config, items):
    import torch
    if not os.environ.get("BD_RUN_SLOW"):
        skip_slow = pytest.mark.skip(reason="slow characterisation run: set BD_RUN_SLOW=1")
        for item in items:
            if "slow" in item.keywords:
                item.add_marker(skip_slow)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
