import os
import sys

import pytest

# the oracle's OpenMP loops are tiny in the tests: a 128-core box spends its time in fork/join
os.environ.setdefault("OMP_NUM_THREADS", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_artifacts():
    """Make sure the HIP library and the CPU oracle are built (hipcc / gcc are in the image; the
    in-tree .so normally travels with the repo snapshot, this only covers a fresh checkout)."""
    from polysolve_amd import _lib, build as _b
    if not os.path.exists(_lib.LIB_PATH):
        _b.build()
    import oracle as O
    O.build()


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
