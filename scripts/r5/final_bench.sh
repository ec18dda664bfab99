#!/bin/bash
R=${GRAFT_REPO_ROOT:-.}
cd $R
timeout 900 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
python - <<'P'
import json
j = json.loads(open("gpurun_out/r05_bench.json").read().strip().splitlines()[-1])
r = j["roofline"]; e = j["elasticity"]; h = j["host_contract"]; ns = j["north_star"]
print("value", round(j["value"] / 1e6, 2), "M DOF/s; ms/it", round(j["ms_per_iteration"], 4), "pat", round(r["avg_launch_ms"], 4), round(r["frac"], 4),
      "csr", round(r["csr_plain"]["avg_launch_ms"], 4), round(r["csr_plain"]["frac"], 4), "it_frac", round(j["iteration_roofline"]["fused_frac_of_peak"], 4))
print("unstructured", {k: (round(v["frac"], 3), round(v["dof_per_s"] / 1e6, 1)) for k, v in r["unstructured"].items()})
print("elasticity solve", round(e["solve_s"] * 1e3, 1), e["iterations"], "setup", round(e["generate_plus_setup_s"], 3), "refresh", round(e["generate_plus_refresh_s"], 3), "spmv", round(e["spmv"]["frac"], 3))
u = e["unstructured"]["random_nodes"]; print("  shuffled", round(u["solve_s"] * 1e3, 1), u["iterations"], "caller", round(u["caller_numbering"]["solve_s"] * 1e3, 1), u["caller_numbering"]["iterations"])
for k in ("poisson", "elasticity"):
    print("host", k, {kk: round(v["seconds"], 4) for kk, v in h[k].items() if isinstance(v, dict) and "seconds" in v})
print("north_star", {k: (round(v["setup_s"], 3), round(v["solve_s"], 4), v["iterations"]) for k, v in ns.items() if isinstance(v, dict) and "setup_s" in v})
print("cpu", round(j["cpu_baseline"]["value"] / 1e6, 3), "probe", j["box"]["probe"])
P
tail -2 gpurun_out/r05_bench.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
