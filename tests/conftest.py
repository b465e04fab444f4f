import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts: build them once (no-op otherwise)
    import __graft_entry__
    __graft_entry__.ensure_built()


@pytest.fixture(scope="session")
def cases():
    with open(os.path.join(GOLDEN, "cases.json")) as fh:
        return json.load(fh)


def load_fixture(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def fixture_loader():
    return load_fixture
