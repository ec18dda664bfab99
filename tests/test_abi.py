"""CPU checks of the drop-in boundary: libpsolve_hip.so loads, exports every symbol that
include/psolve_hip.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from polysolve_amd import _lib, build
    build.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "psolve_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psolve_hip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from polysolve_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/psolve_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "python binding and header disagree"
    assert lib.psolve_hip_abi_version() == 2


def test_info_struct_layout_matches_header():
    from polysolve_amd import _lib
    text = open(os.path.join(ROOT, "include", "psolve_hip.h")).read()
    body = re.search(r"typedef struct psolve_hip_info \{(.*?)\} psolve_hip_info;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(int64_t|int32_t|double)\s+(\w+);", body)
    ctype = {"int64_t": C.c_int64, "int32_t": C.c_int32, "double": C.c_double}
    assert [(n, ctype[t]) for t, n in fields] == list(_lib.Info._fields_)


def test_no_link_time_dependency_on_rccl_or_torch():
    import subprocess
    from polysolve_amd import _lib
    out = subprocess.run(["objdump", "-p", _lib.LIB_PATH], capture_output=True, text=True).stdout
    needed = re.findall(r"NEEDED\s+(\S+)", out)
    assert any("amdhip64" in n for n in needed)
    assert not any("rccl" in n or "torch" in n or "c10" in n or "oracle" in n for n in needed), needed


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_fails_loudly_without_gpu(lib):
    from polysolve_amd import Solver
    h = C.c_void_p()
    rc = lib.psolve_hip_create(C.byref(h), 0)
    assert rc == -2 and not h.value  # PSOLVE_HIP_EDEVICE
    assert b"device" in lib.psolve_hip_last_error(None).lower()
    with pytest.raises(RuntimeError):
        Solver.create("HIP", "")


def test_factory_contract():
    from polysolve_amd import Solver
    assert Solver.available_solvers() == ["HIP"]
    with pytest.raises(RuntimeError, match="Unrecognized solver type"):  # Solver.cpp:495
        Solver.create("NoSuchSolver", "")


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "polysolve_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "libpsolve_oracle" not in text and "oracle/_" not in text, f


def test_plan_halo_host_only(lib):
    from polysolve_amd import plan_halo
    offs = [0, 10, 20, 30]
    halo, counts = plan_halo(1, 3, offs, [3, 12, 25, 3, 9, 29, 15, 10, 19])
    assert halo.tolist() == [3, 9, 25, 29] and counts.tolist() == [2, 0, 2]
    halo, counts = plan_halo(0, 1, [0, 5], [0, 1, 4])
    assert halo.size == 0 and counts.tolist() == [0]
    with pytest.raises(RuntimeError):
        plan_halo(0, 2, [0, 5, 10], [11])  # outside the global matrix
    # 7-point slab: halo = one z-plane from each neighbour
    import oracle as O
    nx = ny = 4
    nz = 6
    offs = [0, 2 * 16, 4 * 16, 6 * 16]
    A = O.poisson7(nx, ny, nz, 2, 4)
    halo, counts = plan_halo(1, 3, offs, A.col)
    assert halo.tolist() == list(range(16, 32)) + list(range(64, 80))
    assert counts.tolist() == [16, 0, 16]


def test_cpp_host_example_builds_and_fails_loudly_without_gpu(tmp_path):
    """examples/solve_mm.cpp: a compiled (C++17, g++) host of the C ABI with the reference's
    loadSymmetric Matrix-Market reader; on a box without a GPU it must stop in psolve_hip_create."""
    import subprocess
    from polysolve_amd import build
    build.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    exe = os.path.join(ROOT, "examples", "solve_mm")
    assert os.path.exists(exe)
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present: covered by tests/test_gpu_solver.py::test_cpp_host_matrix_market")
    mtx = tmp_path / "t.mtx"
    mtx.write_text("%%MatrixMarket matrix coordinate real symmetric\n2 2 2\n1 1 2.0\n2 2 3.0\n")
    p = subprocess.run([exe, str(mtx)], capture_output=True, text=True)
    assert p.returncode == 2 and "psolve_hip_create failed" in p.stderr


def test_row_partition_of_the_multi_device_handle(lib):
    """psolve_hip_partition_rows (host-only, the split psolve_hip_factorize makes on a multi-device handle):
    contiguous cover, balanced by stored entries, cuts at block_size multiples, refusal of tiny matrices."""
    import numpy as np
    import oracle as O
    for A, world, bs in ((O.poisson7(12, 10, 16), 2, 1), (O.poisson7(9, 8, 14), 3, 1), (O.elasticity_q1(8), 4, 3),
                         (O.gr_30_30(), 8, 1)):
        off = np.zeros(world + 1, np.int64)
        rp = np.ascontiguousarray(A.rowptr, np.int32)
        assert lib.psolve_hip_partition_rows(A.n, rp.ctypes.data, world, bs, off.ctypes.data) == 0
        assert off[0] == 0 and off[-1] == A.n and np.all(np.diff(off) > 0) and np.all(off % bs == 0)
        per = np.diff(rp[off].astype(np.int64))
        assert per.max() <= 1.3 * A.nnz / world
    off = np.zeros(5, np.int64)
    rp = np.arange(0, 9, dtype=np.int32)
    assert lib.psolve_hip_partition_rows(8, rp.ctypes.data, 4, 1, off.ctypes.data) != 0
    assert b"too small to partition" in lib.psolve_hip_last_error(None)


def test_host_pattern_hash(lib):
    """psolve_hip_host_pattern_hash (host-only; how factorize(host arrays) recognises the pattern it holds on the device):
    independent of the number of threads, sensitive to one changed column id, to two swapped column ids and to a moved
    row boundary, and not a function of the values."""
    import numpy as np
    import oracle as O
    A = O.poisson7(41, 37, 29)  # 44 k rows, 300 k entries: several 64 Ki-entry tiles
    rp, col = np.ascontiguousarray(A.rowptr, np.int32), np.ascontiguousarray(A.col, np.int32)

    def h(rp_, col_, threads):
        out = np.zeros(2, np.uint64)
        assert lib.psolve_hip_host_pattern_hash(len(rp_) - 1, len(col_), rp_.ctypes.data, col_.ctypes.data, threads,
                                                out.ctypes.data) == 0
        return tuple(int(v) for v in out)
    ref = h(rp, col, 1)
    assert all(h(rp, col, t) == ref for t in (0, 2, 3, 5, 8, 16))
    c2 = col.copy()
    c2[123456] += 1
    assert h(rp, c2, 4)[1] != ref[1] and h(rp, c2, 4)[0] == ref[0]
    c3 = col.copy()
    c3[[70000, 70001]] = c3[[70001, 70000]]
    assert h(rp, c3, 4)[1] != ref[1]
    r2 = rp.copy()
    r2[100] += 1
    assert h(r2, col, 4)[0] != ref[0]
    assert lib.psolve_hip_host_pattern_hash(5, 3, None, col.ctypes.data, 1, np.zeros(2, np.uint64).ctypes.data) != 0
    empty = np.zeros(1, np.int32)
    assert h(empty, np.zeros(0, np.int32), 3) == h(empty, np.zeros(0, np.int32), 1)


def test_permutation_is_a_bijection_without_a_gpu():
    """psolve_hip_permutation (the renumbering of the bench's unstructured leg) is host-only: a bijection of [0, n),
    confined to its windows in mode 2, and different for different seeds."""
    import numpy as np
    from polysolve_amd import HIPSolver
    for n, mode, w in [(1, 1, 2), (2, 1, 2), (1000, 1, 2), (4097, 1, 2), (100000, 2, 4096), (12345, 2, 100), (7, 2, 2)]:
        p = HIPSolver.permutation(n, mode, w, seed=7)
        assert np.array_equal(np.sort(p), np.arange(n)), (n, mode, w)
        if mode == 2:
            assert np.array_equal(p // w, np.arange(n) // w)
    a, b = HIPSolver.permutation(5000, 1, 2, seed=7), HIPSolver.permutation(5000, 1, 2, seed=8)
    assert not np.array_equal(a, b) and np.count_nonzero(a == np.arange(5000)) < 50
