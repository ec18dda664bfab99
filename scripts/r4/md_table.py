"""markdown table of a per-level CSV of scripts/amg_by_level.py (levels 0, 1 and the PCG launches): python scripts/r4/md_table.py file.csv [max_level]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
maxl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print("| level | operation | launches / iteration | avg us (live) | MB / launch | GB/s | of 8 TB/s | share | kernel |")
print("|---|---|---|---|---|---|---|---|---|")
for r in rows:
    lv = r["level"]
    if lv.lstrip("-").isdigit() and int(lv) > maxl: continue
    name = f"L{lv}" if lv.lstrip("-").isdigit() and int(lv) >= 0 else "pcg"
    print(f"| {name} | {r['op']} | x{r['launches_per_iteration']} | {float(r['avg_us']):.1f} | {float(r['bytes_per_launch'])/1e6:.1f} | {float(r['gbs']):.0f} | "
          f"{float(r['frac_of_peak']):.3f} | {100*float(r['share_of_iteration']):.1f} % | `{r['kernel']}` |")
