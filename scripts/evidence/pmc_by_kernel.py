"""Mean PMC counter values per kernel from rocprofv3 counter_collection CSVs (one directory per --pmc pass).
usage: pmc_by_kernel.py out.json dir1 [dir2 ...] [--min-grid N]   -- kernels are keyed by their demangled name, shortened;
launches whose grid is below --min-grid threads are dropped (small levels run the same templates)."""
import csv, glob, json, re, sys, collections
args = [a for a in sys.argv[1:] if not a.startswith("--")]
min_grid = 0
for i, a in enumerate(sys.argv):
    if a == "--min-grid": min_grid = int(sys.argv[i + 1]); args.remove(sys.argv[i + 1])
out_path, dirs = args[0], args[1:]
agg = collections.OrderedDict()
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if int(row.get("Grid_Size", 0) or 0) < min_grid: continue
            kn = re.sub(r"\(anonymous namespace\)::|psolve::", "", row["Kernel_Name"]); kn = re.sub(r"^void ", "", kn); kn = re.sub(r"\(.*", "", kn)
            agg.setdefault(kn, collections.OrderedDict()).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
out = {}
for kn, cs in agg.items():
    out[kn] = {}
    for c, v in cs.items():
        out[kn][c] = dict(mean=sum(v) / len(v), max=max(v), n=len(v))
json.dump(out, open(out_path, "w"), indent=1)
for kn, cs in out.items():
    print(kn[:100])
    for c, s in cs.items(): print(f"    {c:38s} mean={s['mean']:.6g} max={s['max']:.6g} n={s['n']}")
