#!/bin/bash
# round 3: the whole GPU suite, then the bench line as the driver runs it
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --durations=6 2>&1 | tail -18
timeout 900 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
tail -c 300 gpurun_out/r03_bench.err
python - <<'PY'
import json
j = json.loads([l for l in open('gpurun_out/r03_bench.json') if l.startswith('{')][-1])
r = j["roofline"]
print("value", j["value"], "ms/it", j["ms_per_iteration"], "its", j["iterations"])
print("pat  ", r["avg_launch_ms"], r["frac"])
print("csr  ", r["csr_plain"]["avg_launch_ms"], r["csr_plain"]["frac"], r["csr_plain"]["dof_per_s"])
for k, u in r["unstructured"].items(): print("unstr", k, u["avg_launch_ms"], u["frac"], u["dof_per_s"], u["iterations"])
e = j["elasticity"]; print("elast", e["solve_s"], e["iterations"], e["generate_plus_setup_s"], e["generate_plus_refresh_s"], e["spmv"]["avg_launch_ms"], e["spmv"]["frac"])
print("cpu  ", j["cpu_baseline"])
ns = j["north_star"]
for k in ("gpu_reference_config", "gpu_recommended_config", "cpu_amgcl_single_socket"): print(k, {a: b for a, b in ns[k].items() if a not in ("amg", "what")})
PY
