import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_detect():
    return load_golden("detect.json")


@pytest.fixture(scope="session")
def golden_camshift():
    return load_golden("camshift.json")


@pytest.fixture(scope="session")
def golden_facetrackr():
    return load_golden("facetrackr.json")


@pytest.fixture(scope="session")
def cascade():
    from headtrackr_amd.cascade import load_cascade

    return load_cascade()
