#!/bin/bash
# kernel trace of full AMG setups on one handle (scripts/evidence/setup_laps.py): the kernels of the LAST setup in time order + totals
R=${GRAFT_REPO_ROOT:-.}
cd /tmp && export TMPDIR=/tmp
TAG=${TAG:-default}
D=$R/gpurun_out/${RND:-r06}_prof_setup_$TAG; rm -rf $D
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -o s -- python $R/scripts/evidence/setup_laps.py > $R/gpurun_out/${RND:-r06}_prof_setup_$TAG.log 2>&1
T=$(find $D -name "*kernel_trace*" | head -1)
python - "$T" $R/gpurun_out/${RND:-r06}_setup_${TAG}_by_kernel.csv <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('psolve::', '').split('(')[0][:60]
starts = [i for i, r in enumerate(rows) if 'poisson7_kernel' in r['Kernel_Name'] or 'elasticity_fill_kernel' in r['Kernel_Name']]
ks = rows[starts[-1]:]
t0 = int(ks[0]['Start_Timestamp'])
span = (max(int(r['End_Timestamp']) for r in ks) - t0) / 1e6
agg = collections.OrderedDict()
for r in ks:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault(nm(r), [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
print(f"{len(ks)} launches in the last setup, span {span:.2f} ms, kernel time {tot/1e3:.2f} ms")
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f); w.writerow(["kernel", "launches", "us", "share_of_kernel_time"])
    for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, "%.1f" % us, "%.4f" % (us / tot)])
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {us:9.1f} us {100*us/tot:5.1f} %  calls={c:4d}  {k}")
P
rm -rf $D
