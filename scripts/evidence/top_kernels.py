"""Print the top rows of a rocprofv3 kernel_stats CSV with short names."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:k]:
    name = re.sub(r"\(anonymous namespace\)::|psolve::", "", r["Name"])
    name = re.sub(r"\(.*", "", name)
    print(f"{int(r['TotalDurationNs'])/1e6:9.3f} ms  calls={int(r['Calls']):5d}  avg={float(r['AverageNs'])/1e3:9.1f} us  {name[:70]}")
