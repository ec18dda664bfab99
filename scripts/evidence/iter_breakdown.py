"""Per-iteration kernel breakdown of a solve from a rocprofv3 kernel_stats CSV: python iter_breakdown.py CSV ITERATIONS"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
its = float(sys.argv[2])
setup = ("agg_", "rowset", "spgemm", "prolongation", "graph_check", "col_scatter", "rowsort", "strength", "scan", "hash",
         "elasticity", "poisson7", "transpose", "extract", "gershgorin", "fill", "count", "pat_", "sell_", "expand",
         "diag_inverse", "col_hist", "rowlen", "gather_kernel", "splitmix", "bound", "block_values", "block_power")
tot = 0.0
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|psolve::", "", r["Name"])
    name = re.sub(r"\(.*", "", name)
    if any(k in name for k in setup):
        continue
    t = int(r["TotalDurationNs"]) / 1e6
    tot += t
    if t / its >= 0.01:
        print(f"{t/its:7.3f} ms/it {int(r['Calls'])/its:6.1f} calls/it avg {float(r['AverageNs'])/1e3:7.1f} us  {name[:70]}")
print("sum", round(tot / its, 3), "ms per iteration")
