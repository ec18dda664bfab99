"""The JSON half of the factory: what ``Solver::create(const json&, logger, strict)`` does before it reaches
the backend (/root/reference/src/polysolve/linear/Solver.cpp:74-158) -- pick the first available solver of a
priority list, validate the parameters against the spec, inject the defaults -- restated for the one backend
this package provides.  The rules for ``/HIP`` are the file a PolySolve build merges into its own
``linear-solver-spec.json`` (integration/linear-solver-spec.hip.json); the root rules below mirror that
file's ``/``, ``/solver``, ``/precond`` and ``/enable_overwrite_solver`` entries (linear-solver-spec.json:2-67).
Host logic only: no GPU, no native library."""
from __future__ import annotations

import copy
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SPEC_PATH = os.path.join(os.path.dirname(_HERE), "integration", "linear-solver-spec.hip.json")

PRECOND_OPTIONS = ["Eigen::IdentityPreconditioner", "Eigen::DiagonalPreconditioner", "Eigen::IncompleteCholesky",
                   "Eigen::LeastSquareDiagonalPreconditioner", "Eigen::IncompleteLUT"]  # linear-solver-spec.json:55-67


def load_rules(available_solvers=("HIP",), default_solver="HIP", default_precond="Eigen::DiagonalPreconditioner"):
    """Root rules + the /HIP objects, with /solver and /precond filled in from availability the way
    apply_default_solver does at run time (Solver.cpp:74-90)."""
    hip = json.load(open(SPEC_PATH))
    rules = [
        {"pointer": "/", "default": None, "type": "object",
         "optional": ["enable_overwrite_solver", "solver", "precond", "AMGCL"] + hip["append"]["/"]["optional"]},
        # the reference's own block (linear-solver-spec.json:153-454 validates it in a PolySolve build); read by this
        # backend only under /HIP/amgcl_params -- accepted here as it is
        {"pointer": "/AMGCL", "default": None, "type": "object", "opaque": True},
        {"pointer": "/enable_overwrite_solver", "default": False, "type": "bool"},
        {"pointer": "/solver", "default": default_solver, "type": "string", "options": list(available_solvers)},
        {"pointer": "/precond", "default": default_precond, "type": "string", "options": [""] + PRECOND_OPTIONS},
    ]
    return rules + hip["rules"]


def _rule_for(rules, pointer):
    for r in rules:
        if r["pointer"] == pointer:
            return r
    parent, _, _ = pointer.rpartition("/")
    for r in rules:  # list items: "/HIP/devices/*"
        if r["pointer"] == parent + "/*":
            return r
    return None


_TYPES = {
    "object": lambda v: isinstance(v, dict),
    "list": lambda v: isinstance(v, (list, tuple)),
    "string": lambda v: isinstance(v, str),
    "bool": lambda v: isinstance(v, bool),
    "int": lambda v: isinstance(v, int) and not isinstance(v, bool),
    "float": lambda v: isinstance(v, (int, float)) and not isinstance(v, bool),
}


def verify(params, rules, strict=True, pointer="/"):
    """Errors (strings) of `params` against `rules`; strict = unknown keys are errors (jse.strict)."""
    errors = []
    rule = _rule_for(rules, pointer)
    if rule is None:
        return [f"{pointer}: no rule"] if strict else []
    if not _TYPES[rule["type"]](params):
        return [f"{pointer}: expected {rule['type']}, got {type(params).__name__} ({params!r})"]
    if "options" in rule and params not in rule["options"]:
        errors.append(f"{pointer}: {params!r} is not one of {rule['options']}")
    if "min" in rule and params < rule["min"]:
        errors.append(f"{pointer}: {params!r} < min {rule['min']}")
    if "max" in rule and params > rule["max"]:
        errors.append(f"{pointer}: {params!r} > max {rule['max']}")
    base = "" if pointer == "/" else pointer
    if rule.get("opaque"):
        return errors
    if rule["type"] == "object":
        allowed = set(rule.get("optional", [])) | set(rule.get("required", []))
        for key in rule.get("required", []):
            if key not in params:
                errors.append(f"{pointer}: missing required key '{key}'")
        for key, value in params.items():
            if key not in allowed:
                if strict:
                    errors.append(f"{pointer}: unknown key '{key}'")
                continue
            errors += verify(value, rules, strict, f"{base}/{key}")
    elif rule["type"] == "list":
        for i, value in enumerate(params):
            errors += [e.replace(f"{base}/*", f"{base}/{i}") for e in verify(value, rules, strict, f"{base}/*")]
    return errors


def inject_defaults(params, rules, pointer="/"):
    """A copy of `params` with every missing optional key that has a non-null default filled in
    (jse.inject_defaults, Solver.cpp:152).  An absent object whose default is null stays absent -- which is
    why every reference backend guards with `params.contains(name())` (EigenSolver.tpp:70, AMGCL.cpp:108) --
    so the backend's built-in defaults must equal the spec's (tests/test_spec.py checks that)."""
    out = copy.deepcopy(params)
    rule = _rule_for(rules, pointer)
    if rule is None or rule["type"] != "object" or not isinstance(out, dict):
        return out
    base = "" if pointer == "/" else pointer
    for key in rule.get("optional", []):
        child = _rule_for(rules, f"{base}/{key}")
        if child is None:
            continue
        if key in out:
            out[key] = inject_defaults(out[key], rules, f"{base}/{key}")
        elif child.get("default") is not None:
            out[key] = copy.deepcopy(child["default"])
    return out


def select_valid_solver(params, available_solvers, default_solver, warn=None):
    """Solver.cpp:92-134: a list under /solver -> the first available entry ("" if none);
    /enable_overwrite_solver -> an unavailable name falls back to the default."""
    warn = warn or (lambda msg: None)
    if isinstance(params.get("solver"), (list, tuple)):
        accepted = next((s for s in params["solver"] if s in available_solvers), "")
        if not accepted:
            warn("No valid solver found in the list of specified solvers!")
        params["solver"] = accepted
    if params.get("enable_overwrite_solver", False):
        if not isinstance(params.get("solver"), str) or params["solver"] not in available_solvers:
            warn(f"Solver {params.get('solver')} is invalid, falling back to {default_solver}")
            params["solver"] = default_solver
    return params
