#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 300 python bench.py --precond amg --no-cpu-baseline --no-north-star --no-extra --steps 3 --warmup 1 > gpurun_out/r04_bench_amg.json 2> gpurun_out/r04_bench_amg.err
python - <<'P'
import json
j = json.loads(open("gpurun_out/r04_bench_amg.json").read().strip().splitlines()[-1])
print("amg 256^3", round(j["ms_per_step"], 2), "ms", j["iterations"], "its")
for L in j.get("amg_cycle_ops", []):
    print(" level", L["level"], {k: (round(v["us"], 1), round(v["frac_of_peak"], 3)) for k, v in L["ops"].items()})
P
tail -3 gpurun_out/r04_bench_amg.err
python - <<'P'
import json, os, sys
sys.path.insert(0, os.getcwd())
from polysolve_amd import HIPSolver
import bench
e = bench.elasticity_block(HIPSolver, 100)
json.dump(e, open("gpurun_out/r04_elasticity_block.json", "w"), indent=1)
print("elasticity solve", round(e["solve_s"] * 1e3, 1), "ms", e["iterations"], "its; setup", round(e["generate_plus_setup_s"], 3), "refresh", round(e["generate_plus_refresh_s"], 3))
for L in e["cycle_ops"]:
    print(" level", L["level"], {k: (round(v["us"], 1), round(v["frac_of_peak"], 3)) for k, v in L["ops"].items()})
u = e["unstructured"]["random_nodes"]
print("shuffled nodes:", round(u["solve_s"] * 1e3, 1), "ms", u["iterations"], "its; caller numbering", round(u["caller_numbering"]["solve_s"] * 1e3, 1), u["caller_numbering"]["iterations"])
P
