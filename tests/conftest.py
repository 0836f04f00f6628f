import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the tests force kernels, poison arenas and inject failures through wfahip_set_option: debug knobs, refused without this
os.environ.setdefault("WFAHIP_DEBUG", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first(request):
    """On a GPU box, bring up torch's HIP runtime before the library's (bench.py's order): torch ships its own runtime,
    and tests that hold device buffers in torch tensors need it to have seen the GPUs first."""
    if request.config.getoption("-m") and "not gpu" in request.config.getoption("-m"):
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda:0")
    except Exception:
        pass


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
