"""polysolve_amd -- MI355X (gfx950) "HIP" linear-solver backend for PolySolve.

The product is libpsolve_hip.so (polysolve_amd/csrc, C ABI in include/psolve_hip.h); this package is
the thin host-side mirror of the reference's Solver interface for that backend.  It never imports
the CPU oracle and has no CPU fallback.
"""
from .solver import DeviceArray, HIPSolver, HostHierarchy, LocalGroup, ic_host_factorize, Solver, plan_halo  # noqa: F401

__all__ = ["Solver", "HIPSolver", "DeviceArray", "HostHierarchy", "LocalGroup", "plan_halo"]
