"""Per-LEVEL kernel table of an AMG-PCG solve from a rocprofv3 kernel trace (round 4).

    python scripts/evidence/amg_by_level.py <kernel_trace.csv> [--plan plan.json] [--out by_level.csv] [--groups groups.csv]

Two views of the same trace:

* `--groups` (always printed): launches grouped by (kernel, grid size, LDS size); inside every group the NO-OP launches
  -- the iterations queued behind the converged one return at once -- are dropped (duration < 0.3 x the group's
  median) and the live-launch average is reported.  Averages over all calls of a kernel NAME (what `--stats` prints)
  mix levels and no-ops: a level-0 restriction of 190 us and a level-2 one of 21 us show up as "51 us".

* `--plan`: the launch sequence of ONE PCG iteration, written by the driver that set the hierarchy up
  (`iteration_plan()` below restates cycle() / cheb_solve() of polysolve_amd/csrc/amg.hip from the level shapes): every
  entry names its level and operation and carries the ALGORITHMIC bytes of that launch.  The trace is cut into
  iterations by matching the kernel names against the plan's regular expressions position by position, so two launches
  of the same kernel and grid on different levels (P_0 and P_1 both run `spmv_csr_dma<128, ADD>` on 1792 workgroups)
  are told apart by where they sit in the cycle.  Iterations whose PCG product is a no-op are dropped.  Output: one
  row per plan entry -- level, operation, kernel, live calls, average / minimum us, bytes, GB/s, fraction of 8 TB/s,
  share of the iteration -- and the same folded by (level, operation).
"""
from __future__ import annotations

import argparse
import csv
import json
import re
import statistics
import sys

PEAK_GBS = 8000.0
SKIP = re.compile(r"__amd_rocclr_(copyBuffer|fillBuffer)")  # memcpy / memset kernels of the runtime: not part of the plan


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::|psolve::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def load_trace(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append({"name": short(r["Kernel_Name"]), "grid": int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])),
                         "lds": int(r["LDS_Block_Size"]), "vgpr": int(r["VGPR_Count"]),
                         "t0": int(r["Start_Timestamp"]), "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
    rows.sort(key=lambda r: r["t0"])
    return rows


def groups_table(rows):
    g = {}
    for r in rows:
        g.setdefault((r["name"], r["grid"], r["lds"]), []).append(r["us"])
    out = []
    for (name, grid, lds), v in g.items():
        med = statistics.median(v)
        live = [u for u in v if u >= 0.3 * med]
        out.append({"kernel": name, "grid": grid, "lds": lds, "calls": len(v), "live": len(live), "noop": len(v) - len(live),
                    "avg_live_us": sum(live) / len(live), "min_us": min(live), "max_us": max(live),
                    "total_live_ms": sum(live) / 1e3, "avg_all_us": sum(v) / len(v)})
    out.sort(key=lambda r: -r["total_live_ms"])
    return out


# ---- the plan: launch sequence of one PCG iteration ----------------------------------------------------------------
def op_bytes(fmt, rows, cols, nnz, op):
    """algorithmic bytes of one launch.  fmt "bsr3": nnz = stored 3x3 blocks (76 B each, rows / cols scalar counts);
    "csr": 12 B per stored entry.  Per row: 4 B pointer (block rows: 4 B per 3 rows) + the vectors the epilogue moves."""
    if fmt == "bsr3":
        mat = 76 * nnz + 4 * (rows // 3)
    elif fmt == "pat":
        mat = 8 * nnz + 6 * rows  # values + 16-bit pattern id + row pointer
    else:
        mat = 12 * nnz + 4 * rows
    if fmt == "kinds":      # round 5: row kinds / block-row kinds -- a 16-bit kind per (block) row, no matrix stream
        mat = 2 * rows
    if fmt == "bkinds":
        mat = 2 * (rows // 3)
    x_in = 8 * cols
    vec = {"residual": 16 * rows,              # f in, t out
           "restrict": 8 * rows,               # f_c out
           "prolong": 16 * rows,               # x in / out
           "dot": 8 * rows,                    # q out (x = p is the gathered vector)
           "cheb": 40 * rows + (24 * rows if fmt in ("bsr3", "bkinds") else 0),  # f, p in / out, x' out, D^-1 (block: 9 per node)
           }[op]
    return mat + x_in + vec


def iteration_plan(levels, prm):
    """levels: [{n, A: {fmt, rows, cols, nnz}, P: {...} | None, R: {...} | None, fused: bool, block: bool}, ...];
    prm: ncycle, npre, npost, cheb_degree.  Mirrors cycle() / cheb_solve() in amg.hip."""
    plan = []

    def add(level, op, regex, nbytes):
        plan.append({"level": level, "op": op, "re": regex, "bytes": int(nbytes)})

    def cheb_solve(l, zero):
        L = levels[l]
        A = L["A"]
        for k in range(prm["cheb_degree"]):
            if k == 0 and zero:
                if L["block"]:
                    add(l, "cheb_first", r"block_cheb_update_kernel", 8 * (1 + 1 + 1 + 3) * L["n"])
                else:
                    add(l, "cheb_first", r"cheb_first_kernel", 8 * 4 * L["n"])
                continue
            if L["block"] and not L["fused"]:
                add(l, "cheb_residual", r"spmv_", op_bytes(A["fmt"], A["rows"], A["cols"], A["nnz"], "residual"))
                add(l, "cheb_update", r"block_cheb_update_kernel", 8 * (1 + 2 + 2 + 3) * L["n"])
            else:
                add(l, "cheb_step", r"spmv_", op_bytes(A["fmt"], A["rows"], A["cols"], A["nnz"], "cheb"))

    def cycle(l, zero):
        L = levels[l]
        if l + 1 == len(levels):
            # round 5: a relaxed coarsest level of at most amg.coarse_dense rows is one dense product per visit
            if L["n"] <= prm.get("coarse_dense", 1024):
                add(l, "dense_coarse", r"dense_matvec_kernel", 8 * L["n"] * L["n"] + 16 * L["n"])
                return
            for _ in range(prm["npre"] + prm["npost"]):
                cheb_solve(l, zero)
                zero = False
            return
        for _ in range(prm["ncycle"]):
            for _ in range(prm["npre"]):
                cheb_solve(l, zero)
                zero = False
            zero = False
            A, P, R = L["A"], L["P"], L["R"]
            add(l, "residual", r"spmv_", op_bytes(A["fmt"], A["rows"], A["cols"], A["nnz"], "residual"))
            add(l, "restrict", r"spmv_", op_bytes(R["fmt"], R["rows"], R["cols"], R["nnz"], "restrict"))
            cycle(l + 1, True)
            add(l, "prolong", r"spmv_", op_bytes(P["fmt"], P["rows"], P["cols"], P["nnz"], "prolong"))
            for _ in range(prm["npost"]):
                cheb_solve(l, False)

    A0 = levels[0]["A"]
    add(-1, "pcg_product", r"spmv_", op_bytes(A0["fmt"], A0["rows"], A0["cols"], A0["nnz"], "dot"))
    add(-1, "pcg_update_xr", r"pcg_update_xr_kernel", 8 * 5 * levels[0]["n"])
    add(-1, "pcg_check", r"pcg_check_kernel", 0)
    cycle(0, True)
    add(-1, "pcg_dot", r"dot_kernel", 8 * 2 * levels[0]["n"])
    add(-1, "pcg_update_p", r"pcg_update_p_kernel", 8 * 3 * levels[0]["n"])
    return plan


def match_plan(rows, plan):
    seq = [r for r in rows if not SKIP.search(r["name"])]
    res = [re.compile(p["re"]) for p in plan]
    n, m = len(seq), len(plan)
    hits, i = [], 0
    while i + m <= n:
        if res[0].search(seq[i]["name"]) and all(res[j].search(seq[i + j]["name"]) for j in range(1, m)):
            hits.append(seq[i:i + m])
            i += m
        else:
            i += 1
    return hits


def by_level(rows, plan):
    hits = match_plan(rows, plan)
    if not hits:
        return None
    med0 = statistics.median(h[0]["us"] for h in hits)
    live = [h for h in hits if h[0]["us"] >= 0.3 * med0]
    it_us = statistics.mean(sum(r["us"] for r in h) for h in live)
    out = []
    for j, p in enumerate(plan):
        v = [h[j]["us"] for h in live]
        k = live[0][j]
        avg = sum(v) / len(v)
        gbs = p["bytes"] / (avg * 1e-6) / 1e9 if avg > 0 else 0.0
        out.append({"pos": j, "level": p["level"], "op": p["op"], "kernel": k["name"], "grid": k["grid"], "lds": k["lds"],
                    "live_calls": len(v), "avg_us": avg, "min_us": min(v), "bytes": p["bytes"], "gbs": gbs,
                    "frac_of_peak": gbs / PEAK_GBS, "share_of_iteration": avg / it_us})
    return {"iterations_matched": len(hits), "iterations_live": len(live), "iteration_us": it_us, "rows": out}


def fold(rows):
    f = {}
    for r in rows:
        k = (r["level"], r["op"], r["kernel"])
        a = f.setdefault(k, {"level": r["level"], "op": r["op"], "kernel": r["kernel"], "launches_per_iteration": 0, "us_per_iteration": 0.0,
                             "bytes_per_launch": r["bytes"], "share_of_iteration": 0.0})
        a["launches_per_iteration"] += 1
        a["us_per_iteration"] += r["avg_us"]
        a["share_of_iteration"] += r["share_of_iteration"]
    out = []
    for a in f.values():
        a["avg_us"] = a["us_per_iteration"] / a["launches_per_iteration"]
        a["gbs"] = a["bytes_per_launch"] / (a["avg_us"] * 1e-6) / 1e9 if a["avg_us"] > 0 else 0.0
        a["frac_of_peak"] = a["gbs"] / PEAK_GBS
        out.append(a)
    out.sort(key=lambda a: (a["level"] if a["level"] >= 0 else 99, -a["us_per_iteration"]))
    return out


def write_csv(path, rows, cols):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(cols)
        for r in rows:
            w.writerow([("%.4g" % r[c]) if isinstance(r[c], float) else r[c] for c in cols])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--plan")
    ap.add_argument("--out")
    ap.add_argument("--groups")
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    rows = load_trace(a.trace)
    g = groups_table(rows)
    print(f"{len(rows)} launches, {len(g)} (kernel, grid, lds) groups; top by live time:")
    for r in g[:a.top]:
        print(f"  {r['total_live_ms']:9.3f} ms  live={r['live']:5d} noop={r['noop']:4d}  avg_live={r['avg_live_us']:8.1f} us "
              f"(all calls: {r['avg_all_us']:8.1f})  grid={r['grid']:5d} lds={r['lds']:6d}  {r['kernel'][:64]}")
    if a.groups:
        write_csv(a.groups, g, ["kernel", "grid", "lds", "calls", "live", "noop", "avg_live_us", "min_us", "max_us",
                                "total_live_ms", "avg_all_us"])
    if a.plan:
        plan = json.load(open(a.plan))
        res = by_level(rows, plan["plan"] if isinstance(plan, dict) else plan)
        if res is None:
            print("plan: no iteration of the trace matches the launch sequence", file=sys.stderr)
            return 1
        print(f"\nplan of {len(res['rows'])} launches per iteration: {res['iterations_matched']} iterations matched, "
              f"{res['iterations_live']} live; {res['iteration_us'] / 1e3:.3f} ms per live iteration")
        folded = fold(res["rows"])
        for r in folded:
            lvl = "pcg" if r["level"] < 0 else f"L{r['level']}"
            print(f"  {lvl:>4} {r['op']:<14} x{r['launches_per_iteration']:<2d} {r['avg_us']:8.1f} us  {r['bytes_per_launch'] / 1e6:9.1f} MB "
                  f"{r['gbs']:7.0f} GB/s  {r['frac_of_peak']:.3f} of peak  {100 * r['share_of_iteration']:5.1f} % of the iteration  {r['kernel'][:48]}")
        if a.out:
            write_csv(a.out, folded, ["level", "op", "kernel", "launches_per_iteration", "avg_us", "us_per_iteration",
                                      "bytes_per_launch", "gbs", "frac_of_peak", "share_of_iteration"])
            write_csv(a.out.replace(".csv", "_by_position.csv"), res["rows"],
                      ["pos", "level", "op", "kernel", "grid", "lds", "live_calls", "avg_us", "min_us", "bytes", "gbs",
                       "frac_of_peak", "share_of_iteration"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
