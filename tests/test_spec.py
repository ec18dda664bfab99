"""The JSON half of the factory (Solver::create(json, logger, strict), Solver.cpp:74-158) and the `/HIP` spec
artifact a PolySolve build merges into linear-solver-spec.json.  Host logic only."""
import json
import os
import re
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.fixture(scope="module")
def spec():
    from polysolve_amd import spec as S
    return S


def _leaves(rules):
    for r in rules:
        if r["pointer"] == "/HIP/amgcl_params":
            continue  # read by the adapters (HIPSolver.hpp, solver.py), like "devices": not a parameter of the C ABI
        if r["pointer"].startswith("/HIP/") and r["type"] not in ("object", "list") and not r["pointer"].endswith("/*"):
            yield r["pointer"][len("/HIP/"):].replace("/", "."), r


def test_spec_defaults_equal_the_backends_defaults(spec):
    """Every leaf of /HIP is a key psolve_hip_set_param accepts, with the built-in default (inject_defaults leaves
    an absent /HIP object absent, so the two must agree), and every key the backend accepts is in the spec."""
    import ctypes as C
    from polysolve_amd import _lib
    L = _lib.load()
    rules = spec.load_rules()
    seen = set()
    for key, rule in _leaves(rules):
        v = C.c_double()
        assert L.psolve_hip_default_param(key.encode(), C.byref(v)) == 0, key
        seen.add(key)
        d = rule["default"]
        if key == "precond":
            assert d == "" and v.value == 1.0  # "" = keep the factory's choice; the backend's own default: jacobi
        elif rule["type"] == "string":  # /HIP/amg/aggregation ...: names on the JSON side, codes at the C ABI
            from polysolve_amd.solver import AMG_NAMES
            assert key.startswith("amg.") and float(AMG_NAMES[key[4:]][d]) == v.value, key
            assert sorted(rule["options"]) == sorted(AMG_NAMES[key[4:]]), key
        elif key == "tolerance":
            assert d < 0  # alias, "not set"
        elif rule["type"] == "bool":
            assert float(bool(d)) == v.value, key
        else:
            assert float(d) == pytest.approx(v.value, rel=1e-9), key
    src = open(os.path.join(ROOT, "polysolve_amd", "csrc", "solver.hip")).read()
    body = src[src.index("bool param_value("):src.index("double Context::get_param")]
    accepted = set(re.findall(r'k == "([a-z0-9_.]+)"', body))
    assert accepted == seen, accepted ^ seen
    v = C.c_double()
    assert L.psolve_hip_default_param(b"no_such_key", C.byref(v)) != 0


def test_verify_accepts_valid_blocks_and_rejects_the_rest(spec):
    rules = spec.load_rules()
    good = [
        {},
        {"solver": "HIP"},
        {"solver": "HIP", "precond": "Eigen::IdentityPreconditioner", "HIP": {"tolerance": 1e-11, "max_iter": 500}},
        {"solver": "HIP", "HIP": {"precond": "amg", "block_size": 3, "devices": [0, 1, 2, 3],
                                  "amg": {"ncycle": 2, "cheb_degree": 16, "matrix_fp32": True, "eps_strong": 0.08}}},
    ]
    for p in good:
        assert spec.verify(p, rules) == [], p
    bad = [
        ({"solver": "Hypre"}, "/solver"),                                   # not available here
        ({"solver": "HIP", "HIP": {"tolerence": 1e-8}}, "unknown key 'tolerence'"),
        ({"solver": "HIP", "HIP": {"amg": {"degree": 3}}}, "unknown key 'degree'"),
        ({"solver": "HIP", "HIP": {"max_iter": "many"}}, "/HIP/max_iter: expected int"),
        ({"solver": "HIP", "HIP": {"max_iter": 2.5}}, "/HIP/max_iter: expected int"),
        ({"solver": "HIP", "HIP": {"true_residual": 1}}, "/HIP/true_residual: expected bool"),
        ({"solver": "HIP", "HIP": {"block_size": 4}}, "/HIP/block_size"),
        ({"solver": "HIP", "HIP": {"precond": "ilu"}}, "/HIP/precond"),
        ({"solver": "HIP", "HIP": {"devices": [0, "one"]}}, "/HIP/devices/1: expected int"),
        ({"solver": "HIP", "HIP": {"devices": 0}}, "/HIP/devices: expected list"),
        ({"solver": "HIP", "HIP": {"relative_tolerance": -1e-3}}, "< min"),
        ({"solver": "HIP", "MAS": {}}, "unknown key 'MAS'"),
        ({"precond": "Eigen::Nope"}, "/precond"),
    ]
    for p, needle in bad:
        errs = spec.verify(p, rules, strict=True)
        assert errs and any(needle in e for e in errs), (p, errs)
    # strict_validation = false: unknown keys pass, wrong types still do not
    assert spec.verify({"solver": "HIP", "HIP": {"tolerence": 1e-8}, "MAS": {}}, rules, strict=False) == []
    assert spec.verify({"solver": "HIP", "HIP": {"max_iter": "many"}}, rules, strict=False)


def test_inject_defaults_and_solver_selection(spec):
    rules = spec.load_rules()
    p = spec.inject_defaults({}, rules)
    assert p["solver"] == "HIP" and p["precond"] == "Eigen::DiagonalPreconditioner" and p["enable_overwrite_solver"] is False
    assert "HIP" not in p  # an absent object with a null default stays absent (EigenSolver.tpp:70 guards for that)
    p = spec.inject_defaults({"solver": "HIP", "HIP": {"max_iter": 7, "amg": {"ncycle": 2}}}, rules)
    assert p["HIP"]["max_iter"] == 7 and p["HIP"]["relative_tolerance"] == 1e-8 and p["HIP"]["devices"] == [0]
    assert p["HIP"]["amg"]["ncycle"] == 2 and p["HIP"]["amg"]["cheb_degree"] == 16 and p["HIP"]["precond"] == ""
    assert spec.verify(p, rules) == []
    msgs = []
    q = spec.select_valid_solver({"solver": ["Hypre", "HIP", "AMGCL"]}, ["HIP"], "HIP", msgs.append)
    assert q["solver"] == "HIP" and not msgs
    q = spec.select_valid_solver({"solver": ["Hypre", "Pardiso"]}, ["HIP"], "HIP", msgs.append)
    assert q["solver"] == "" and msgs
    q = spec.select_valid_solver({"solver": "Pardiso", "enable_overwrite_solver": True}, ["HIP"], "HIP", msgs.append)
    assert q["solver"] == "HIP"


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (this container only)")
def test_hooks_apply_to_the_reference_tree(tmp_path):
    """integration/apply_hip_hooks.py on a scratch copy of the four upstream files it touches: the six hooks
    land next to the MAS ones, the merged spec still parses, knows /HIP and lists "HIP" where the factory
    looks (root optional list, /solver options); applying twice changes nothing."""
    import importlib.util
    for rel in ("CMakeLists.txt", "linear-solver-spec.json", "src/polysolve/linear/Solver.cpp",
                "src/polysolve/linear/CMakeLists.txt"):
        os.makedirs(os.path.dirname(tmp_path / rel), exist_ok=True)
        shutil.copy(os.path.join(REF, rel), tmp_path / rel)
    sp = importlib.util.spec_from_file_location("apply_hip_hooks", os.path.join(ROOT, "integration", "apply_hip_hooks.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    mod.apply(str(tmp_path))
    first = {rel: open(tmp_path / rel).read() for rel in ("CMakeLists.txt", "linear-solver-spec.json",
                                                           "src/polysolve/linear/Solver.cpp",
                                                           "src/polysolve/linear/CMakeLists.txt")}
    mod.apply(str(tmp_path))
    for rel, text in first.items():
        assert open(tmp_path / rel).read() == text
    cpp = first["src/polysolve/linear/Solver.cpp"]
    assert cpp.count("POLYSOLVE_WITH_HIP") == 3 and '#include "HIPSolver.hpp"' in cpp
    assert 'else if (solver == "HIP")' in cpp and "std::make_unique<HIPSolver>(precond)" in cpp
    assert cpp.index('"MAS",') < cpp.index('"HIP",')
    rules = json.loads(first["linear-solver-spec.json"])
    root = next(r for r in rules if r["pointer"] == "/")
    assert "HIP" in root["optional"] and "MAS" in root["optional"]
    assert "HIP" in next(r for r in rules if r["pointer"] == "/solver")["options"]
    ptrs = [r["pointer"] for r in rules]
    assert "/HIP" in ptrs and "/HIP/amg/cheb_degree" in ptrs and len(ptrs) == len(set(ptrs))
    assert "option(POLYSOLVE_WITH_HIP" in first["CMakeLists.txt"] and "libpsolve_hip.so" in first["CMakeLists.txt"]
    assert "HIPSolver.hpp" in first["src/polysolve/linear/CMakeLists.txt"]
    # the merged rules validate a HIP block with this package's validator, and still reject a typo
    from polysolve_amd import spec as S
    merged = [r for r in rules if r["pointer"] in ("/", "/solver", "/precond", "/enable_overwrite_solver")
              or r["pointer"].startswith("/HIP")]
    assert S.verify({"solver": "HIP", "HIP": {"precond": "amg", "amg": {"ncycle": 2}}}, merged) == []
    assert S.verify({"solver": "HIP", "HIP": {"precnd": "amg"}}, merged)


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("cmake") is None, reason="needs the reference checkout and cmake")
def test_patched_linear_cmakelists_configures(tmp_path):
    """The hook in src/polysolve/linear/CMakeLists.txt must survive a real configure with POLYSOLVE_WITH_HIP=ON: an
    out-of-tree header inside ${SOURCES} makes source_group(TREE ...) abort (round-2 advice).  A scratch project
    declares the polysolve_linear target, then add_subdirectory()s a patched copy of the reference's directory
    (its source files present, nothing compiled: configure + generate only)."""
    import importlib.util
    import subprocess
    lin = tmp_path / "src" / "polysolve" / "linear"
    shutil.copytree(os.path.join(REF, "src", "polysolve", "linear"), lin)
    for rel in ("CMakeLists.txt", "linear-solver-spec.json"):
        shutil.copy(os.path.join(REF, rel), tmp_path / rel)
    sp = importlib.util.spec_from_file_location("apply_hip_hooks", os.path.join(ROOT, "integration", "apply_hip_hooks.py"))
    mod = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(mod)
    mod.apply(str(tmp_path))
    top = tmp_path / "scratch"
    top.mkdir()
    (top / "CMakeLists.txt").write_text(
        "cmake_minimum_required(VERSION 3.18)\nproject(hook_check LANGUAGES CXX)\n"
        "option(POLYSOLVE_WITH_HIP \"\" ON)\nset(POLYSOLVE_WITH_CUDA OFF)\n"
        f"set(PSOLVE_HIP_ROOT \"{ROOT}\")\n"
        "add_library(polysolve_linear STATIC)\n"
        "set_target_properties(polysolve_linear PROPERTIES LINKER_LANGUAGE CXX)\n"
        f"add_subdirectory(\"{lin}\" linear)\n")
    out = subprocess.run(["cmake", "-S", str(top), "-B", str(top / "build")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "is not a prefix of file" not in out.stderr


def test_amgcl_block_translation():
    """/HIP/amgcl_params: the reference's /AMGCL block (AMGCL.cpp:32-128: defaults patched by the caller's objects) as
    /HIP keys -- host logic, no GPU."""
    from polysolve_amd.solver import HIPSolver
    d = HIPSolver.amgcl_block_to_hip({})
    assert d == {"precond": "amg", "tolerance": 1e-10, "max_iter": 1000,
                 "amg": {"max_levels": 6, "ncycle": 2, "coarsening": "smoothed_aggregation", "relax_type": "chebyshev",
                         "direct_coarse": False, "cheb_degree": 16, "cheb_power_iters": 100, "cheb_higher": 2,
                         "cheb_lower": 0.008333333333, "cheb_scale": True, "sa_relax": 1, "estimate_spectral_radius": True,
                         "eps_strong": 0}}
    d = HIPSolver.amgcl_block_to_hip({"AMGCL": {"block_size": 3, "solver": {"tol": 1e-8, "maxiter": 50},
                                                "precond": {"ncycle": 1, "npre": 2, "relax": {"degree": 4},
                                                            "coarsening": {"aggr": {"eps_strong": 0.08}, "relax": 0.9}}}})
    assert d["tolerance"] == 1e-8 and d["max_iter"] == 50 and d["block_size"] == 3
    assert d["amg"]["ncycle"] == 1 and d["amg"]["cheb_degree"] == 4 and d["amg"]["cheb_power_iters"] == 100
    assert d["amg"]["eps_strong"] == 0.08 and d["amg"]["sa_relax"] == 0.9 and d["amg"]["npre"] == 2 and "npost" not in d["amg"]
    # round 5: amgcl's other runtime classes (linear-solver-spec.json:393-397, 423-427; AMGCL.cpp:67-92) are translated too
    d = HIPSolver.amgcl_block_to_hip({"AMGCL": {"precond": {"direct_coarse": True, "relax": {"type": "damped_jacobi", "damping": 0.6},
                                                            "coarsening": {"type": "aggregation", "over_interp": 1.8}}}})
    assert d["amg"]["relax_type"] == "damped_jacobi" and d["amg"]["damping"] == 0.6 and d["amg"]["direct_coarse"] is True
    assert d["amg"]["coarsening"] == "aggregation" and d["amg"]["over_interp"] == 1.8 and "cheb_degree" not in d["amg"]
    d = HIPSolver.amgcl_block_to_hip({"AMGCL": {"precond": {"relax": {"type": "spai0"}}}})
    assert d["amg"]["relax_type"] == "spai0" and d["amg"]["coarsening"] == "smoothed_aggregation"
    d = HIPSolver.amgcl_block_to_hip({"AMGCL": {"precond": {"relax": {"scale": False}}}})
    assert d["amg"]["cheb_scale"] is False
    # round 6: the ordered relaxations and the class "relaxation"
    d = HIPSolver.amgcl_block_to_hip({"AMGCL": {"precond": {"class": "relaxation", "relax": {"type": "ilu0", "damping": 0.9}}}})
    assert d["amg"]["class"] == "relaxation" and d["amg"]["relax_type"] == "ilu0" and d["amg"]["ilu_damping"] == 0.9
    d = HIPSolver.amgcl_block_to_hip({"AMGCL": {"precond": {"relax": {"type": "gauss_seidel"}}}})
    assert d["amg"]["relax_type"] == "gauss_seidel" and "class" not in d["amg"] and "cheb_degree" not in d["amg"]
    for bad in ({"solver": {"type": "bicgstab"}}, {"precond": {"class": "nested"}},
                {"precond": {"relax": {"type": "spai1"}}}, {"precond": {"coarsening": {"type": "ruge_stuben"}}}):
        with pytest.raises(RuntimeError):
            HIPSolver.amgcl_block_to_hip({"AMGCL": bad})
    from polysolve_amd import spec as sp
    rules = sp.load_rules()
    assert sp.verify({"solver": "HIP", "HIP": {"amgcl_params": True}, "AMGCL": {"precond": {"ncycle": 1}}}, rules) == []
