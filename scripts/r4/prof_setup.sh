#!/bin/bash
# kernel trace of full AMG setups on one handle (scripts/r4/setup_second.py): the kernels of the LAST setup in time order
R=${GRAFT_REPO_ROOT:-.}
cd /tmp && export TMPDIR=/tmp
D=$R/gpurun_out/r04_prof_setup; rm -rf $D
N=${N:-216} timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -o s -- python $R/scripts/r4/setup_second.py > $R/gpurun_out/r04_prof_setup.log 2>&1
T=$(find $D -name "*kernel_trace*" | head -1)
python - "$T" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(r): return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').replace('psolve::', '').split('(')[0][:46]
# setups start with poisson7_kernel
starts = [i for i, r in enumerate(rows) if 'poisson7_kernel' in r['Kernel_Name'] or 'elasticity_fill_kernel' in r['Kernel_Name']]
ks = rows[starts[-1]:]
t0 = int(ks[0]['Start_Timestamp'])
prev_end = t0
print(f"{len(ks)} launches in the last setup, span {(int(ks[-1]['End_Timestamp'])-t0)/1e6:.2f} ms")
for r in ks:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3
    d = (e - s) / 1e3
    if d > float(__import__("os").environ.get("MINUS", "250")) or gap > 400:
        print(f"{(s-t0)/1e6:8.2f} ms  {d:9.1f} us  gap {gap:8.1f} us  grid {r['Grid_Size_X']:>9}  {nm(r)}")
    prev_end = max(prev_end, e)
P
