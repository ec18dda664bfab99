import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SETS"] = '{};{"refresh_power_iters":4};{"refresh_power_iters":0}'
exec(open(os.path.join(ROOT, "scripts", "r5", "ab.py")).read())
