"""Per-kernel table of ONE numeric refresh: the kernel stats of a run with 1 + K2 refreshes minus those of a run with 1 + K1
refreshes, divided by K2 - K1.  usage: refresh_table.py stats_K1.csv stats_K2.csv K1 K2 [out.csv]"""
import csv, re, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        name = re.sub(r"\(anonymous namespace\)::|psolve::", "", r["Name"]); name = re.sub(r"^void ", "", name); name = re.sub(r"\(.*", "", name)
        d[name] = (int(r["Calls"]), int(r["TotalDurationNs"]))
    return d
a, b, k1, k2 = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rows = []
for name, (cb, tb) in b.items():
    ca, ta = a.get(name, (0, 0))
    dc, dt = (cb - ca) / (k2 - k1), (tb - ta) / (k2 - k1)
    if dc > 0: rows.append((dt / 1e3, dc, name))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"one refresh: {tot / 1e3:.2f} ms of kernel time in {sum(r[1] for r in rows):.0f} launches")
for us, c, name in rows[:int(sys.argv[6]) if len(sys.argv) > 6 else 30]:
    print(f"  {us:9.1f} us  {100 * us / tot:5.1f} %  calls={c:6.1f}  avg={us / c:8.1f} us  {name[:90]}")
if len(sys.argv) > 5:
    with open(sys.argv[5], "w", newline="") as f:
        w = csv.writer(f); w.writerow(["kernel", "launches_per_refresh", "us_per_refresh", "avg_us", "share"])
        for us, c, name in rows: w.writerow([name, "%.1f" % c, "%.1f" % us, "%.1f" % (us / c), "%.4f" % (us / tot)])
