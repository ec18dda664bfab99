"""polysolve_amd/host/HIPSolver.hpp -- the `class HIPSolver : public polysolve::linear::Solver` a PolySolve build
registers as Solver::create("HIP") -- compiled against the interface stand-in of tests/stubs/ (Eigen, nlohmann
and the reference headers are not in the image) and, on the GPU box, driven through the reference's call sequence."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = ["-I" + os.path.join(ROOT, p) for p in ("tests/stubs", "include", "polysolve_amd/host")]
DRIVER = os.path.join(ROOT, "tests", "adapter_driver.cpp")
EXE = os.path.join(ROOT, "tests", "_build", "adapter_driver")


# An nlohmann/json single header that happens to be in the image (3.1.1, shipped with another package under /opt/conda):
# where it exists the adapter is ALSO compiled and driven against the real json class (round-4 review, missing #5: until
# round 5 it had only ever met the stand-in of tests/stubs).  Eigen and the reference's own headers remain stand-ins.
REAL_JSON = next((p for p in ("/opt/conda/include/json.hpp", "/usr/include/nlohmann/json.hpp", "/usr/local/include/nlohmann/json.hpp")
                  if os.path.exists(p)), None)


def _build(large_index: bool = False, real_json: bool = False):
    from polysolve_amd import _lib
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    if real_json:
        exe = EXE + "_real_json"
        libdir = os.path.dirname(_lib.LIB_PATH)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-DPSOLVE_TEST_REAL_NLOHMANN=\"%s\"" % REAL_JSON, *INC,
                               DRIVER, "-o", exe, "-L" + libdir, "-lpsolve_hip", "-Wl,-rpath," + libdir])
        return exe
    if large_index:  # the reference's POLYSOLVE_LARGE_INDEX build (Types.hpp:11-15): 64-bit indices at the boundary
        exe = EXE + "_large_index"
        libdir = os.path.dirname(_lib.LIB_PATH)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-DPOLYSOLVE_LARGE_INDEX", *INC, DRIVER, "-o", exe,
                               "-L" + libdir, "-lpsolve_hip", "-Wl,-rpath," + libdir])
        return exe
    if (not os.path.exists(EXE) or os.path.getmtime(EXE) < max(
            os.path.getmtime(DRIVER), os.path.getmtime(os.path.join(ROOT, "polysolve_amd/host/HIPSolver.hpp")),
            os.path.getmtime(os.path.join(ROOT, "tests/stubs/polysolve/linear/Solver.hpp")))):
        libdir = os.path.dirname(_lib.LIB_PATH)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", *INC, DRIVER, "-o", EXE, "-L" + libdir,
                               "-lpsolve_hip", "-Wl,-rpath," + libdir])
    return EXE


def test_adapter_compiles_and_links():
    """Syntax + link: every C entry point the adapter calls exists in libpsolve_hip.so with that signature."""
    _build()
    assert os.path.exists(EXE)
    assert os.path.exists(_build(large_index=True))  # std::ptrdiff_t indices: narrowed by the adapter


@pytest.mark.skipif(REAL_JSON is None, reason="no nlohmann/json header in this image")
def test_adapter_compiles_against_the_real_nlohmann_json():
    """The same adapter and driver with polysolve::json = nlohmann::json of the header found in the image: every JSON call
    the adapter makes (count, iterators with key() / value(), get<T>, operator[] on checked keys) exists there with the
    semantics it relies on -- compiled -Wall -Werror and linked."""
    assert os.path.exists(_build(real_json=True))


@pytest.mark.gpu
@pytest.mark.parametrize("shards,large_index", [(1, False), (3, False), (1, True)])
def test_adapter_runs_the_reference_call_sequence(shards, large_index):
    exe = _build(large_index)
    out = subprocess.run([exe, str(shards)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ADAPTER_OK" in out.stdout
    assert "warning: preconditioner 'Eigen::IncompleteLUT'" in out.stderr


@pytest.mark.gpu
@pytest.mark.skipif(REAL_JSON is None, reason="no nlohmann/json header in this image")
@pytest.mark.parametrize("shards", [1, 3])
def test_adapter_runs_with_the_real_nlohmann_json(shards):
    """... and RUN through the reference's call sequence with real nlohmann::json objects for params and info."""
    exe = _build(real_json=True)
    out = subprocess.run([exe, str(shards)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ADAPTER_OK" in out.stdout
