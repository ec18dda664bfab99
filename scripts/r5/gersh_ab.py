"""Recommended configuration with the smoothers' radii by power iterations (20) against Gershgorin bounds (cheb_power_iters 0,
cheb_higher 1.0: the bound needs no safety factor): refresh, setup, solve.  env KIND, N"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SETS"] = '{};{"cheb_power_iters":0,"cheb_higher":1.0};{"cheb_power_iters":0,"cheb_higher":1.0,"cheb_lower":0.08}'
exec(open(os.path.join(ROOT, "scripts", "r5", "ab.py")).read())
