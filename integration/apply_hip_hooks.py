#!/usr/bin/env python3
"""Wire the "HIP" backend into a PolySolve checkout:  python integration/apply_hip_hooks.py <polysolve-root>

Edits the checkout in place by ANCHORS (regular expressions on the lines the MAS backend is wired with), so
this file carries none of the upstream text and keeps working when line numbers move.  Idempotent.

  1. src/polysolve/linear/Solver.cpp   include of HIPSolver.hpp next to the MAS include   (Solver.cpp:62-64)
  2.                                   factory branch `else if (solver == "HIP")`          (Solver.cpp:400-405)
  3.                                   "HIP" in available_solvers()                       (Solver.cpp:538-540)
  4. linear-solver-spec.json           "HIP" in the root `optional` list and in `/solver` `options`, and the
                                       `/HIP` rules of integration/linear-solver-spec.hip.json appended
  5. src/polysolve/linear/CMakeLists.txt   the header-only adapter as a source of polysolve_linear (after, not in, the
                                       ${SOURCES} list that source_group(TREE ...) walks)
  6. CMakeLists.txt                    option POLYSOLVE_WITH_HIP, definition, include dirs, link to libpsolve_hip.so
"""
from __future__ import annotations

import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
MARK = "POLYSOLVE_WITH_HIP"


def _insert_after(text: str, anchor: str, block: str, what: str, count_from_end: bool = False) -> str:
    """Insert `block` after the first line group matching `anchor` (a regex that ends at a line end)."""
    m = list(re.finditer(anchor, text, flags=re.M))
    if not m:
        raise SystemExit(f"apply_hip_hooks: anchor for {what} not found")
    at = (m[-1] if count_from_end else m[0]).end()
    return text[:at] + block + text[at:]


def patch_solver_cpp(path: str) -> None:
    s = open(path).read()
    if MARK in s:
        return
    # (1) include, right after the MAS include group
    s = _insert_after(s, r'#ifdef POLYSOLVE_WITH_MAS\n#include "MASSolver\.hpp"\n#endif\n',
                      '#ifdef POLYSOLVE_WITH_HIP\n#include "HIPSolver.hpp"\n#endif\n', "the include")
    # (2) factory branch, right after the MAS branch (same brace-straddling style as its neighbours)
    s = _insert_after(s, r'#ifdef POLYSOLVE_WITH_MAS\n\s*\}\n\s*else if \(solver == "MAS"\)\n\s*\{\n[^\n]*\n#endif\n',
                      '#ifdef POLYSOLVE_WITH_HIP\n        }\n        else if (solver == "HIP")\n        {\n'
                      '            return std::make_unique<HIPSolver>(precond);\n#endif\n', "the factory branch")
    # (3) available_solvers(), right after the MAS entry
    s = _insert_after(s, r'#ifdef POLYSOLVE_WITH_MAS\n\s*"MAS",\n#endif\n',
                      '#ifdef POLYSOLVE_WITH_HIP\n            "HIP",\n#endif\n', "available_solvers()")
    open(path, "w").write(s)


def patch_spec(path: str) -> None:
    rules = json.load(open(path))
    hip = json.load(open(os.path.join(HERE, "linear-solver-spec.hip.json")))
    if any(r.get("pointer") == "/HIP" for r in rules):
        return
    for r in rules:
        extra = hip["append"].get(r.get("pointer"))
        if extra:
            for key, values in extra.items():
                r.setdefault(key, [])
                r[key] += [v for v in values if v not in r[key]]
    rules += hip["rules"]
    json.dump(rules, open(path, "w"), indent=4)
    open(path, "a").write("\n")


def patch_linear_cmake(path: str) -> None:
    s = open(path).read()
    if MARK in s:
        return
    # NOT appended to ${SOURCES}: that list feeds source_group(TREE "${CMAKE_CURRENT_SOURCE_DIR}" ...), which aborts
    # the configure step for a file outside the tree ("ROOT ... is not a prefix of file").  The adapter is
    # header-only and reachable through the include directories the root CMakeLists adds; it is listed as a
    # source of the target only so that IDEs show it, AFTER the source_group / target_sources pair.
    block = ('\nif(POLYSOLVE_WITH_HIP)\n    # header-only adapter (outside this tree); the kernels are prebuilt in libpsolve_hip.so\n'
             '    target_sources(polysolve_linear PRIVATE ${PSOLVE_HIP_ROOT}/polysolve_amd/host/HIPSolver.hpp)\nendif()\n')
    s = _insert_after(s, r'^target_sources\(polysolve_linear PRIVATE \$\{SOURCES\}\)\n', block, "the target's source list")
    open(path, "w").write(s)


def patch_root_cmake(path: str) -> None:
    s = open(path).read()
    if MARK in s:
        return
    opt = 'option(POLYSOLVE_WITH_HIP          "Enable the MI355X HIP PCG backend (libpsolve_hip.so)" OFF)\n'
    s = _insert_after(s, r'^option\(POLYSOLVE_WITH_CUDA[^\n]*\n', opt, "the option list")
    block = ('\n# MI355X HIP backend: no enable_language(HIP) -- hipcc built libpsolve_hip.so beforehand\n'
             '# (polysolve_amd/csrc/Makefile, --offload-arch=gfx950); PolySolve only compiles the adapter header.\n'
             'if(POLYSOLVE_WITH_HIP)\n'
             '    set(PSOLVE_HIP_ROOT "" CACHE PATH "root of the polysolve_amd repository")\n'
             '    target_compile_definitions(polysolve PUBLIC POLYSOLVE_WITH_HIP)\n'
             '    target_compile_definitions(polysolve_linear PUBLIC POLYSOLVE_WITH_HIP)\n'
             '    target_include_directories(polysolve_linear PUBLIC ${PSOLVE_HIP_ROOT}/include ${PSOLVE_HIP_ROOT}/polysolve_amd/host)\n'
             '    target_link_libraries(polysolve_linear PUBLIC ${PSOLVE_HIP_ROOT}/polysolve_amd/lib/libpsolve_hip.so)\n'
             'endif()\n')
    s = _insert_after(s, r'^\s*target_compile_definitions\(polysolve_linear PUBLIC POLYSOLVE_WITH_MAS\)\n(?:[^\n]*\n)*?endif\(\)\n',
                      block, "the MAS definitions")
    open(path, "w").write(s)


def apply(root: str) -> None:
    patch_solver_cpp(os.path.join(root, "src", "polysolve", "linear", "Solver.cpp"))
    patch_spec(os.path.join(root, "linear-solver-spec.json"))
    patch_linear_cmake(os.path.join(root, "src", "polysolve", "linear", "CMakeLists.txt"))
    patch_root_cmake(os.path.join(root, "CMakeLists.txt"))


if __name__ == "__main__":
    if len(sys.argv) != 2:
        raise SystemExit(__doc__)
    apply(sys.argv[1])
    print("HIP hooks applied to", sys.argv[1])
