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


def _build(large_index: bool = False):
    from polysolve_amd import _lib
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
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


@pytest.mark.gpu
@pytest.mark.parametrize("shards,large_index", [(1, False), (3, False), (1, True)])
def test_adapter_runs_the_reference_call_sequence(shards, large_index):
    exe = _build(large_index)
    out = subprocess.run([exe, str(shards)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ADAPTER_OK" in out.stdout
    assert "warning: preconditioner 'Eigen::IncompleteLUT'" in out.stderr
