import os
import sys
from pathlib import Path

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")  # ordered hipGraph memset nodes (see pl_modules/data_parallel.py)

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def tb():
    from __graft_entry__ import load_package

    return load_package()


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(autouse=True)
def _release_gpu_objects_between_tests():
    """Rollout engines own hipGraphs and private memory pools: collect them right after the test that made them, with the
    device idle, instead of whenever a later capture's gc.collect() happens to run."""
    yield
    import gc

    gc.collect()
    try:
        import torch

        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
            gc.collect()
    except Exception:  # pragma: no cover - CPU-only runs
        pass
