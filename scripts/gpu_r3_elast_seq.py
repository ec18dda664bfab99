"""Why does the SECOND elasticity leg of the bench (nodes renumbered randomly, default) report a slow first setup?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from polysolve_amd import HIPSolver
r, _, _ = bench.elasticity_leg(HIPSolver, 100, 0, 2); print("grid", r["generate_plus_setup_s"], r["generate_plus_refresh_s"], flush=True)
os.environ["PSOLVE_TIMING"] = "1"
r, _, _ = bench.elasticity_leg(HIPSolver, 100, 1, 2); print("random default", r["generate_plus_setup_s"], r["generate_plus_refresh_s"], flush=True)
r, _, _ = bench.elasticity_leg(HIPSolver, 100, 1, 0); print("random caller", r["generate_plus_setup_s"], r["generate_plus_refresh_s"], flush=True)
