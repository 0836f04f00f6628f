import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def known_answers():
    return load_golden("known_answers.json")


@pytest.fixture(scope="session")
def ref_pairs():
    return load_golden("ref_test_pairs.json")


@pytest.fixture(scope="session")
def oracle_vectors():
    return load_golden("oracle_vectors.json")


@pytest.fixture(scope="session")
def built():
    """Build libwfahip.so + the oracle if needed (no-op when the prebuilt files travelled)."""
    import __graft_entry__ as g
    g.build()
    return True
