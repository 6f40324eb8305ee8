import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle32():
    import numpy as np
    from oracle.surfel_oracle import SurfelOracle, build
    build()
    return SurfelOracle(np.float32)


@pytest.fixture(scope="session")
def oracle64():
    import numpy as np
    from oracle.surfel_oracle import SurfelOracle, build
    build()
    return SurfelOracle(np.float64)
