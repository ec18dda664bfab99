"""The detail legs of bench.py (everything that is NOT the one timed region of the bench line): the same system on the
other storages, the bench matrix under pseudo-random renumberings, BASELINE.json configs[2] (block-3 AMG-PCG elasticity),
the host contract PolyFEM / Newton call, the north_star's 10 M-DOF AMG comparison, the state of the box.  bench.py writes
what these return to bench_detail.json and prints a few of their scalars on the bench line (`also`)."""
from __future__ import annotations

import os
import time

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# The AMG configuration this backend recommends (the reference's AMGCL configuration -- W-cycle, Chebyshev-16, 100 power
# iterations, AMGCL.cpp:32-65 -- is timed next to it where it matters): V-cycle, Chebyshev degree 2 on [0.1, 1.1] x the
# power-iteration estimate of rho(D^-1 A), prolongation smoothing over-relaxed by 1.3 (profiles/r03_amg.md)
# Round 6: a factorize of the SAME pattern (Newton's refactorize) continues the smoothers' power iterations from the vector the
# previous factorize ended with for 8 steps instead of 20 from amgcl's random vector ("amg.refresh_power_iters"; first factorizes
# are untouched): iteration counts within 1 of the cold estimate at every step of a ten-step Newton-like sequence
# (tests/test_gpu_amg.py::test_newton_sequence_with_warm_started_refresh), 12 level-0 products less per refresh.
AMG_RECOMMENDED = dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_higher=1.1, cheb_power_iters=20, sa_relax=1.3,
                       refresh_power_iters=8)


# ---- the box this run landed on (round 4): clocks, power, partition modes -------------------------------------------
# gpurun boxes differ (the same binary: level-1 product 177 us on one box, 284 us on another); every number this file
# prints therefore carries the state of the device it was measured on: compute / memory partition mode, power cap,
# DPM level tables, and sclk / mclk / socket power SAMPLED WHILE THE TIMED REGION RUNS (sysfs hwmon, ~50 Hz, a thread).
def _gpu_sysfs(index=0):
    import glob
    cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    if not cards:
        return None, None
    dev = os.path.dirname(cards[min(index, len(cards) - 1)])
    hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
    return dev, (hw[0] if hw else None)


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def box_static(index=0):
    dev, hw = _gpu_sysfs(index)
    out = {"sysfs": dev}
    if not dev:
        return out
    for k in ("current_compute_partition", "current_memory_partition", "power_dpm_force_performance_level"):
        out[k] = _read(os.path.join(dev, k))
    for k in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk"):
        v = _read(os.path.join(dev, k))
        out[k] = v.replace("\n", " | ") if v else None
    if hw:
        for k in ("power1_cap", "power1_cap_default"):
            v = _read(os.path.join(hw, k))
            out[k + "_w"] = int(v) / 1e6 if v and v.isdigit() else None
    out["host_cpus"] = os.cpu_count()
    return out


class BoxSampler:
    """sclk / mclk (MHz), socket power (W), hotspot / memory temperature (C) while a region runs: min / median / max."""
    FILES = {"sclk_mhz": ("freq1_input", 1e-6), "mclk_mhz": ("freq2_input", 1e-6), "power_w": ("power1_input", 1e-6),
             "temp_hotspot_c": ("temp2_input", 1e-3), "temp_mem_c": ("temp3_input", 1e-3)}

    def __init__(self, index=0, period_s=0.02):
        self.dev, self.hw = _gpu_sysfs(index)
        self.period = period_s
        self.samples = {k: [] for k in self.FILES}
        self.fclk = []
        self._stop = False
        self._th = None

    def _loop(self):
        while not self._stop:
            for k, (f, scale) in self.FILES.items():
                v = _read(os.path.join(self.hw, f))
                if v and v.lstrip("-").isdigit():
                    self.samples[k].append(int(v) * scale)
            v = _read(os.path.join(self.dev, "pp_dpm_fclk"))
            if v:
                for line in v.splitlines():
                    if line.rstrip().endswith("*"):
                        try:
                            self.fclk.append(float(line.split(":")[1].lower().replace("mhz", "").replace("*", "")))
                        except (IndexError, ValueError):
                            pass
            time.sleep(self.period)

    def __enter__(self):
        if self.hw:
            import threading
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._th:
            self._th.join()

    def summary(self):
        def mmm(v):
            if not v:
                return None
            w = sorted(v)
            return {"min": round(w[0], 1), "median": round(w[len(w) // 2], 1), "max": round(w[-1], 1), "samples": len(w)}
        out = {k: mmm(v) for k, v in self.samples.items()}
        out["fclk_mhz"] = mmm(self.fclk)
        return out


def time_solves(s, b, x, n, reps=1, warm_iters=0):
    """`reps` full solves from x0 = 0 with the in-loop SpMV sampled by HIP events; returns (seconds per solve,
    iterations, avg SpMV ms, samples, info)."""
    if warm_iters:
        keep = s.get_param("max_iter")
        s.set_parameters({"HIP": {"max_iter": warm_iters}})
        s.axpby_device(n, 0.0, b, 0.0, x)
        s.solve_device(b, x)
        s.set_parameters({"HIP": {"max_iter": int(keep)}})
    s.synchronize()
    ms, samples, its = 0.0, 0, 0
    t = time.perf_counter()
    for _ in range(reps):
        s.axpby_device(n, 0.0, b, 0.0, x)
        s.solve_device(b, x)
        i = s.info_struct()
        ms += i.spmv_ms_avg * i.spmv_samples
        samples += i.spmv_samples
        its = i.num_iterations
    s.synchronize()
    dt = (time.perf_counter() - t) / reps
    return dt, int(its), ms / max(samples, 1), int(samples), s.get_info()


def spmv_stream_bytes(kernel, n, nnz, npat, nkinds):
    """The bytes the product kernel's storage format streams per launch, and a description of the format"""
    if kernel.startswith(("spmv_csr_kind", "spmv_csr_slots")):
        # rows that repeat pattern AND values (a constant-coefficient grid): a 16-bit row kind per row, the kinds' offsets and
        # values in LDS -- no matrix stream.  x once (the rest of its gathers hit the caches), y once, the kinds
        return 18 * n, ("CSR with row kinds: %d (pattern, values) kinds, 16-bit id per row, no matrix stream "
                        "(2 n + 16 n bytes: kinds, x, y)" % nkinds)
    if kernel.startswith("spmv_csr_pat"):
        # the operator repeats a few column-offset patterns (a 7-point grid: 27): the product reads a 16-bit
        # pattern id per row instead of a 32-bit column per entry -- same columns, same order, same sums
        return 8 * nnz + 22 * n, ("CSR with a pattern dictionary: %d column-offset patterns, 16-bit id per row, no "
                                  "column stream (8 nnz + 22 n bytes)" % npat)
    return 12 * nnz + 20 * n, "CSR (12 nnz + 20 n bytes)"


def spmv_leg(kernel, bytes_per_launch, avg_ms, samples, extra=None):
    gbs = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    out = {"kernel": kernel, "bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_ms, "launches_sampled": samples,
           "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    if extra:
        out.update(extra)
    return out


def unstructured_block(HIPSolver, N):
    """The bench matrix under symmetric pseudo-random renumberings (generated on the device, B = Pi A Pi^T, sorted
    columns): no column-offset pattern repeats, so no dictionary -- the plain 12-byte-per-entry CSR stream with real
    gathers.  "windowed": rows shuffled inside windows of 4096 rows (the locality a mesh numbering keeps);
    "random": one permutation of all rows (every gather its own cache line: the worst case).  Each is solved twice:
    as the backend runs it by default ("reorder" 2: a scattered numbering is renumbered at factorize by a
    Cuthill-McKee search on the device; the search and the permuted copy are timed), and in the caller's numbering
    ("reorder" 0: `caller_numbering`)."""
    out = {}
    kern = "spmv_csr_dma<256, SPMV_DOT, double, nt>" if 8 * N ** 3 >= (96 << 20) else "spmv_csr_pipe<256, SPMV_DOT, double>"
    for name, mode in (("windowed_4096", 2), ("random", 1)):
        legs = {}
        for reorder in (2, 0):
            s = HIPSolver("")
            s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "profile_spmv": 8, "reorder": reorder}})
            s.generate_poisson7_permuted(N, N, N, mode=mode, window=4096, seed=7)
            s.synchronize()
            n, nnz, _ = s.matrix_shape()
            b, x = s.device_array(n), s.device_array(n)
            s.generate_rhs(42, b)
            dt, its, ms, smp, info = time_solves(s, b, x, n, reps=1, warm_iters=32)
            c16 = bool(s.get_param("col16_active"))  # ("spmv_col16": 10 instead of 12 bytes per entry; off by default)
            leg = spmv_leg(s.last_spmv_kernel() or kern, (10 if c16 else 12) * nnz + 20 * n, ms, smp,
                           {"patterns": int(s.get_param("spmv_patterns")), "iterations": its, "solve_s": dt,
                            "dof_per_s": n / dt, "ms_per_iteration": dt * 1e3 / max(its, 1),
                            "true_residual": info["true_residual"], "reordered": bool(s.get_param("reorder.active"))})
            if reorder:
                leg["reorder"] = {"first_factorize_search_plus_copy_s": s.get_param("reorder.seconds"),
                                  "bfs_levels": int(s.get_param("reorder.levels")),
                                  "gather_spread_before": s.get_param("reorder.spread_before"),
                                  "gather_spread_after": s.get_param("reorder.spread_after")}
                t = time.perf_counter()
                s.generate_poisson7_permuted(N, N, N, mode=mode, window=4096, seed=7)  # same pattern: the order is kept
                s.synchronize()
                leg["reorder"]["refactorize_generate_plus_copy_s"] = time.perf_counter() - t
            legs[reorder] = leg
            b.free()
            x.free()
            del s
        out[name] = legs[2]
        out[name]["caller_numbering"] = legs[0]
    return out


def amg_cycle_ops(s, nlevels, block, nnzb0=0, max_level=1, fp32=False):
    """HIP-event time of every operation of the V-cycle on levels 0..max_level, launched on the hierarchy's own operators
    (psolve_hip_amg_time_level_ops), against its algorithmic bytes: 76 B per 3x3 block (block hierarchies) / 12 B per
    stored entry + 4 B per row pointer + the vectors the launch reads and writes (profiles/r04_amg.md has the same table
    from a rocprofv3 trace of the solve itself)."""
    out = []
    for l in range(min(nlevels, max_level + 1)):
        t = s.amg_time_level_ops(l, 10)
        rows, cols, nnz = s.amg_level_matrix_shape(l, 0)
        if block:
            # (round 5: level 0 of a constant-coefficient block operator runs from block-row kinds -- 2 bytes per node, no
            # matrix stream)
            bk = l == 0 and s.get_param("bsr3_row_kinds") > 0
            mat = 2 * (rows // 3) if bk else (40 if fp32 else 76) * (nnzb0 if (l == 0 and nnzb0) else nnz // 9) + 4 * (rows // 3)
        else:
            rk = l == 0 and s.get_param("spmv_row_kinds") > 0  # (... and of a scalar one from row kinds: 2 bytes per row)
            mat = 2 * rows if rk else (8 * nnz + 6 * rows) if (l == 0 and s.get_param("spmv_patterns") > 0) else (12 * nnz + 4 * rows)
        ops = {"cheb_step": (t["cheb_step_us"], mat + 8 * cols + 40 * rows + (24 * rows if block else 0)),
               "residual": (t["residual_us"], mat + 8 * cols + 16 * rows),
               "cheb_first": (t["cheb_first_us"], (8 * 6 if block else 8 * 4) * rows)}
        if l + 1 < nlevels:
            for name, what, vec in (("restrict", 2, 8), ("prolong", 1, 16)):
                r2, c2, z2 = s.amg_level_matrix_shape(l, what)
                m2 = ((40 if fp32 else 76) * (z2 // 9) + 4 * (r2 // 3)) if block else (12 * z2 + 4 * r2)
                ops[name] = (t[name + "_us"], m2 + 8 * c2 + vec * r2)
        out.append({"level": l, "rows": rows, "stored_entries": nnz,
                    "ops": {k: {"us": us, "bytes": int(b), "gbs": (b / (us * 1e-6) / 1e9) if us > 0 else 0.0,
                                "frac_of_peak": (b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS) if us > 0 else 0.0} for k, (us, b) in ops.items()}})
    return out


def elasticity_leg(HIPSolver, M, mode, reorder, amg_extra=None):
    """One configs[2] run: generation (mode 0: the grid's node numbering; 1: the nodes renumbered pseudo-randomly) + setup,
    numeric refresh, best of three solves."""
    amg = dict(AMG_RECOMMENDED)
    amg.update(amg_extra or {})
    s = HIPSolver("")
    s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "precond": "amg", "block_size": 3, "profile_spmv": 4,
                              "reorder": reorder, "amg": amg}})
    gen = (lambda: s.generate_elasticity_q1(M)) if mode == 0 else (lambda: s.generate_elasticity_q1_permuted(M, mode=mode, seed=7))
    gen()  # warm-up: code objects, first-touch allocations
    s.set_parameters({"HIP": {"amg": {"reuse": False}}})
    s.synchronize()
    t = time.perf_counter()
    gen()
    s.synchronize()
    t_setup = time.perf_counter() - t
    s.set_parameters({"HIP": {"amg": {"reuse": True}}})
    gen()  # (a full setup once more: it is this one that keeps its patterns for reuse)
    s.synchronize()
    t_refresh = 1e30
    for _ in range(3):  # same pattern: the numeric refresh (Newton's case), best of three (the first one still allocates)
        t = time.perf_counter()
        gen()
        s.synchronize()
        t_refresh = min(t_refresh, time.perf_counter() - t)
    refreshed = bool(s.get_param("amg.last_setup_reused"))
    # opt-in (round 5, NOT amgcl's estimate): a refresh that keeps the smoothers' radii of the previous factorize
    # ("amg.refresh_power_iters" 0) -- a third of a refresh is the 20 power iterations per level; reported next to the default
    # ... and the two ends beside the configured refresh (AMG_RECOMMENDED: 8 warm-started steps): amgcl's own estimate again
    # (-1: 20 steps from the random vector, the backend's default) and the previous radii kept (0)
    t_refresh_keep = t_refresh_cold = None
    if not amg_extra:
        try:
            for mode in (-1, 0):
                s.set_parameters({"HIP": {"amg": {"refresh_power_iters": mode}}})
                gen()
                s.synchronize()
                t = time.perf_counter()
                gen()
                s.synchronize()
                if mode < 0:
                    t_refresh_cold = time.perf_counter() - t
                else:
                    t_refresh_keep = time.perf_counter() - t
            s.set_parameters({"HIP": {"amg": {"refresh_power_iters": int(amg.get("refresh_power_iters", -1))}}})
            gen()  # back to the configured mode for the solves below
            s.synchronize()
        except Exception:
            pass
    n, nnz, _ = s.matrix_shape()
    b, x = s.device_array(n), s.device_array(n)
    s.generate_rhs(42, b)
    best, its, ms, smp, info = 1e30, 0, 0.0, 0, None
    with BoxSampler() as box:
        for _ in range(3):
            dt, its, ms1, smp1, info = time_solves(s, b, x, n)
            if dt < best:
                best, ms, smp = dt, ms1, smp1
    nb, nnzb = int(s.get_param("bsr3_nb")), int(s.get_param("bsr3_nnzb"))
    levels = [s.amg_level_info(l)[:2] for l in range(int(info["amg_levels"]))]
    cycle_ops = amg_cycle_ops(s, int(info["amg_levels"]), block=True, nnzb0=nnzb, fp32=bool(amg.get("matrix_fp32")))
    out = {"generate_plus_setup_s": t_setup, "generate_plus_refresh_s": t_refresh, "refresh_reused_patterns": refreshed,
           "generate_plus_refresh_keep_radii_s": t_refresh_keep, "generate_plus_refresh_cold_estimate_s": t_refresh_cold,
           "solve_s": best, "iterations": its,
           "dof_per_s": n / best, "ms_per_iteration": best * 1e3 / max(its, 1), "true_residual": info["true_residual"],
           "levels": levels, "amg": amg, "reordered": bool(s.get_param("reorder.active")), "box_during_solves": box.summary(),
           "cycle_ops": cycle_ops,
           "spmv": spmv_leg(s.last_spmv_kernel() or "spmv_bsr3_dma",
                            50 * nb if (s.last_spmv_kernel() or "").startswith("spmv_bsr3_kind") else 76 * nnzb + 52 * nb, ms, smp,
                            {"block_rows": nb, "blocks": nnzb, "block_row_kinds": int(s.get_param("bsr3_row_kinds")),
                             "distinct_blocks": int(s.get_param("bsr3_kind_blocks"))})}
    if out["reordered"]:
        out["reorder"] = {"search_plus_copy_s": s.get_param("reorder.seconds"), "bfs_levels": int(s.get_param("reorder.levels")),
                          "gather_spread_before": s.get_param("reorder.spread_before"),
                          "gather_spread_after": s.get_param("reorder.spread_after")}
    b.free()
    x.free()
    return out, n, nnz


def elasticity_block(HIPSolver, M=100):
    """BASELINE.json configs[2]: 3-D linear elasticity (Q1 hexahedra on an M^3-node cube, one face clamped), 3 M^3 DOF,
    block-3 Chebyshev-smoothed-aggregation AMG PCG (the AMGCL_Block<3> path) -- setup and solve timed separately,
    the in-loop BSR-3 product against its 76 nnzb + 52 nb bytes.  `unstructured`: the same stiffness matrix with its
    nodes renumbered pseudo-randomly (what an unstructured mesh's numbering does to it), as the backend runs it by
    default (renumbered at factorize on the node graph, "reorder" 2) and in the caller's numbering ("reorder" 0)."""
    out, n, nnz = elasticity_leg(HIPSolver, M, 0, 2)
    out = dict({"workload": f"Q1 linear elasticity, {M}^3 nodes, {n} DOF, {nnz} stored entries, block-3 AMG-PCG to "
                            f"||r||/||b||<1e-8, x0=0 (BASELINE.json configs[2])"}, **out)
    try:  # round 5: the coarsest level solved instead of relaxed (/AMGCL/precond/direct_coarse; dense inverse on the device)
        d, _, _ = elasticity_leg(HIPSolver, M, 0, 2, {"direct_coarse": True})
        out["direct_coarse"] = {k: d[k] for k in ("generate_plus_setup_s", "generate_plus_refresh_s", "solve_s", "iterations",
                                                  "dof_per_s", "ms_per_iteration", "true_residual", "levels", "amg")}
    except Exception as e:
        out["direct_coarse"] = {"failed": str(e)}
    try:
        u, _, _ = elasticity_leg(HIPSolver, M, 1, 2)
        u["caller_numbering"], _, _ = elasticity_leg(HIPSolver, M, 1, 0)
        out["unstructured"] = {"random_nodes": u}
        # round 6, OPT-IN configurations on the same randomly numbered matrix (none of them is the reference's arithmetic; the
        # default above is): "amg.matrix_fp32" -- the cycle's operator copies hold single-precision values (40 instead of 76
        # bytes per 3 x 3 block), PCG's own product, every vector and every sum stay double, the stopping test is the same
        # double-precision residual; "amg.aggregation" compact + a directly solved coarsest level (profiles/r06_aggregation.md)
        keep = ("generate_plus_setup_s", "generate_plus_refresh_s", "solve_s", "iterations", "dof_per_s", "ms_per_iteration",
                "true_residual", "levels", "amg", "cycle_ops")
        for name, extra in (("matrix_fp32", {"matrix_fp32": True}),
                            ("compact_direct", {"aggregation": "compact", "direct_coarse": True}),
                            ("compact_direct_matrix_fp32", {"aggregation": "compact", "direct_coarse": True, "matrix_fp32": True})):
            try:
                d, _, _ = elasticity_leg(HIPSolver, M, 1, 2, extra)
                u["opt_in_" + name] = {k: d[k] for k in keep}
            except Exception as e:
                u["opt_in_" + name] = {"failed": str(e)}
    except Exception as e:  # never take the structured numbers down
        out["unstructured"] = {"failed": str(e)}
    return out


def host_contract_leg(HIPSolver, np, kind, size, params):
    """The HOST contract PolyFEM / Newton call on one system of config size (analyze_pattern + factorize + solve on
    host arrays, Newton.cpp:189-211).  The system is generated on the device, copied back once
    (psolve_hip_matrix_copy) and handed over as host arrays: first factorize, two factorizes of the same pattern with
    new values (Newton's refactorize), a solve with host b / x.  Seconds are wall time inside the C entry points
    (psolve_hip_info), bytes over PCIe from "stats.h2d_bytes"."""
    import scipy.sparse as sp
    g = HIPSolver("")
    g.set_parameters({"HIP": {"reorder": 0, "block_size": 3 if kind == "elasticity" else 1}})
    (g.generate_poisson7 if kind == "poisson" else g.generate_elasticity_q1)(size)
    n = g.matrix_shape()[0]
    bd = g.device_array(n)
    g.generate_rhs(42, bd)
    ptr, col, val = g.matrix_to_host()
    b = bd.download()
    bd.free()
    del g
    M = sp.csr_matrix((val, col, ptr), shape=(n, n))
    M.has_canonical_format = True  # (generated with sorted, unique columns: spare the mirror's O(nnz) check)
    nnz = M.nnz
    s = HIPSolver("")
    s.set_parameters({"HIP": params})
    out = {"workload": f"{kind} {size}", "n": n, "nnz": nnz, "matrix_gb": (12 * nnz + 4 * (n + 1)) / 1e9,
           "values_gb": 8 * nnz / 1e9}

    def call(name, f, key):
        h0 = s.get_param("stats.h2d_bytes")
        t = time.perf_counter()
        f()
        wall = time.perf_counter() - t
        moved = s.get_param("stats.h2d_bytes") - h0
        sec = s.get_info()[key]
        out[name] = {"seconds": sec, "wall_with_python_s": wall, "h2d_gb": moved / 1e9, "h2d_gbs_over_the_call": moved / 1e9 / sec if sec > 0 else None}

    call("analyze_pattern", lambda: s.analyze_pattern(M, n), "time_analyze")
    call("factorize_first", lambda: s.factorize(M), "time_factorize")
    M2 = sp.csr_matrix((val * 1.0625, col, ptr), shape=(n, n))  # same pattern, new values (a Newton step's Hessian)
    M2.has_canonical_format = True
    call("factorize_same_pattern", lambda: s.factorize(M2), "time_factorize")
    call("factorize_same_pattern_again", lambda: s.factorize(M), "time_factorize")
    x = np.zeros(n)
    s.solve(b, x)
    x[:] = 0
    call("solve", lambda: s.solve(b, x), "time_solve")
    i = s.get_info()
    out["solve"].update(device_part_s=i["time_solve_device"], iterations=int(i["num_iterations"]), true_residual=i["true_residual"])
    out["pattern_uploads"] = int(s.get_param("stats.pattern_uploads"))
    if params.get("precond") == "amg":
        out["amg_refreshed_on_same_pattern"] = bool(s.get_param("amg.last_setup_reused"))
    return out


def host_contract_block(HIPSolver, np, N=256, M=100):
    out = {}
    for name, kind, size, prm in (("poisson", "poisson", N, dict(tolerance=1e-8, max_iter=20000)),
                                  ("elasticity", "elasticity", M, dict(tolerance=1e-8, precond="amg", block_size=3, amg=dict(AMG_RECOMMENDED)))):
        try:
            out[name] = host_contract_leg(HIPSolver, np, kind, size, prm)
        except Exception as e:
            out[name] = {"failed": str(e)}
    return out


def north_star_block(HIPSolver, np, N=216, run_cpu_leg=None):
    """The north_star's own comparison, inside the bench line: 10 M-DOF 3-D Poisson (N = 216) to 1e-8 on one GPU --
    AMG-PCG, setup (factorize: hierarchy built on the device) and solve timed separately, in the reference's AMGCL
    configuration and in the V-cycle configuration this backend recommends -- next to the CPU restatement of the
    reference's AMGCL path on ONE socket."""
    out = {"workload": f"3-D 7-point Poisson {N}^3 ({N ** 3} DOF), AMG-PCG to ||r||/||b||<1e-8, x0=0"}

    def gpu(amg):
        s = HIPSolver("")
        s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "precond": "amg", "amg": amg}})
        s.generate_poisson7(N)  # warm-up of generator + setup kernels (first-touch allocations, code objects)
        t = time.perf_counter()
        s.set_parameters({"HIP": {"amg": {"reuse": False}}})
        s.generate_poisson7(N)
        s.synchronize()
        t_setup = time.perf_counter() - t  # generation (a few ms on the device) + full hierarchy setup
        # Newton's refactorize (same pattern): the default refresh, and the opt-in one that keeps the smoothers' radii
        t_refresh = t_refresh_keep = None
        try:
            s.set_parameters({"HIP": {"amg": {"reuse": True}}})
            s.generate_poisson7(N)
            s.synchronize()
            t = time.perf_counter()
            s.generate_poisson7(N)
            s.synchronize()
            t_refresh = time.perf_counter() - t
            s.set_parameters({"HIP": {"amg": {"refresh_power_iters": 0}}})
            s.generate_poisson7(N)
            s.synchronize()
            t = time.perf_counter()
            s.generate_poisson7(N)
            s.synchronize()
            t_refresh_keep = time.perf_counter() - t
            s.set_parameters({"HIP": {"amg": {"refresh_power_iters": int(amg.get("refresh_power_iters", -1))}}})
            s.generate_poisson7(N)
            s.synchronize()
        except Exception:
            pass
        n = s.matrix_shape()[0]
        b, x = s.device_array(n), s.device_array(n)
        s.generate_rhs(42, b)
        for _ in range(2):
            s.axpby_device(n, 0.0, b, 0.0, x)
            s.synchronize()
            t = time.perf_counter()
            s.solve_device(b, x)
            t_solve = time.perf_counter() - t
        i = s.get_info()
        return {"setup_s": t_setup, "refresh_s": t_refresh, "refresh_keep_radii_s": t_refresh_keep, "solve_s": t_solve,
                "iterations": int(i["num_iterations"]),
                "true_residual": i["true_residual"], "levels": int(i["amg_levels"]), "dof_per_s": n / t_solve, "amg": amg}

    out["gpu_reference_config"] = gpu(dict(ncycle=2, cheb_degree=16, cheb_power_iters=100))
    out["gpu_recommended_config"] = gpu(dict(AMG_RECOMMENDED))
    # round 5, opt-in: the aggregates by a distance-2 independent set in a dozen parallel rounds instead of AMGCL's sequential
    # sweep ("amg.aggregation" = "parallel": NOT the reference's hierarchy; same iteration count on this operator)
    try:
        out["gpu_recommended_config_parallel_aggregation"] = gpu(dict(AMG_RECOMMENDED, aggregation="parallel"))
    except Exception as e:
        out["gpu_recommended_config_parallel_aggregation"] = {"failed": str(e)}
    if run_cpu_leg is None:  # (bench.py hands its CPU-child launcher over; None: GPU legs only)
        return out
    try:
        cpu = run_cpu_leg("amgcl", grid=N)
        out["cpu_amgcl_single_socket"] = cpu
        ct = cpu["setup_s"] + cpu["solve_s"]
        for k in ("gpu_reference_config", "gpu_recommended_config", "gpu_recommended_config_parallel_aggregation"):
            g = out[k]
            if "solve_s" not in g:
                continue
            g["speedup_solve"] = cpu["solve_s"] / g["solve_s"]
            g["speedup_setup_plus_solve"] = ct / (g["setup_s"] + g["solve_s"])
    except Exception as e:  # never take the GPU numbers down
        out["cpu_amgcl_single_socket"] = {"failed": str(e)}
    return out
