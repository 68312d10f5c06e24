import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    here = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(here, "reference_outputs.json")) as f:
        meta = json.load(f)
    arrays = dict(np.load(os.path.join(here, "frames.npz"), allow_pickle=False))
    return meta, arrays
