import os
import sys

import pytest

# the oracle's OpenMP loops are tiny in the tests: a 128-core box spends its time in fork/join
os.environ.setdefault("OMP_NUM_THREADS", "8")
# RCCL legs of the tests run on ONE node: the bootstrap needs the loopback interface only and no InfiniBand probing (on a
# box whose hostname does not resolve, interface discovery has been seen to take minutes).  Process-level defaults of the
# test processes -- the library itself never touches the environment of its host application.
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_IB_DISABLE", "1")
# the ordered relaxations (amg_sweep.hip) end a sweep that makes no progress after 20 s by publishing NaNs; the tests' systems
# sweep in milliseconds, so a broken sweep shall fail its test within seconds instead of holding the GPU box for minutes
os.environ.setdefault("PSOLVE_SWEEP_LIMIT_MS", "1500")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "rccl_multi: runs RCCL (or peer-mapped collectives) between >= 2 distinct GPUs; skipped on one-GPU boxes")


# the tests that exercise RCCL with more than one rank can only run on a >= 2-GPU box and skip silently
# elsewhere; the summary says how many of them actually executed, so that a GPU test record is self-describing.
_RCCL_MULTI = {"selected": 0, "executed": 0}


def pytest_collection_finish(session):  # (after -m / -k deselection)
    _RCCL_MULTI["selected"] = sum(1 for it in session.items if it.get_closest_marker("rccl_multi"))


def pytest_runtest_logreport(report):
    if report.when == "call" and report.passed and "rccl_multi" in report.keywords:
        _RCCL_MULTI["executed"] += 1


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if _RCCL_MULTI["selected"]:
        terminalreporter.write_line(f"rccl_world_gt1_tests_executed: {_RCCL_MULTI['executed']} of {_RCCL_MULTI['selected']} "
                                    "(the others need >= 2 visible GPUs: RCCL between distinct devices ran only in those)")


@pytest.fixture(scope="session", autouse=True)
def _native_artifacts():
    """Make sure the HIP library and the CPU oracle are built (hipcc / gcc are in the image; the
    in-tree .so normally travels with the repo snapshot, this only covers a fresh checkout)."""
    from polysolve_amd import _lib, build as _b
    if not os.path.exists(_lib.LIB_PATH):
        _b.build()
    import oracle as O
    O.build()


@pytest.fixture(scope="session", autouse=True)
def _poisoned_recycled_blocks():
    """PSOLVE_TEST_POISON=1 (a test-process switch, GPU box only): every device block a handle takes back out of its cache of
    released blocks is filled with 0xFF bytes first ("lab.alloc_cache_poison"), for the whole session -- the suite then proves
    that nothing reads an allocation before writing it."""
    if os.environ.get("PSOLVE_TEST_POISON") == "1":
        # (round 6: the knob belongs to a handle; every handle created from here on takes its default from the environment)
        os.environ["PSOLVE_ALLOC_CACHE_POISON"] = "1"
    yield


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
