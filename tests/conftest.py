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
