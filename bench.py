"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid 256]

metric  : DOF/s to 1e-8 relative residual on 3-D 7-point Poisson (SPD), Jacobi-PCG
workload: BASELINE.json configs[1] -- N=256^3 (16.8 M DOF) on one MI355X; with --gpus N the SAME
          system is row-partitioned (z-slabs) over N ranks, one process per GPU ("scaling": "strong",
          the north_star's 8-GPU target is a strong-scaling one).
step    : one full solve (x0 = 0 -> ||r||/||b|| < 1e-8) with matrix, b and x resident in HBM.
value   : n_global * steps / wall time of the K timed solves (max over ranks).
roofline: the dominant kernel of the timed solves -- PCG's SpMV -- on the bytes THAT kernel streams per launch
          (pattern dictionary on this structured grid: 8*nnz + 22*n; plain CSR: 12*nnz + 20*n) / its HIP-event
          duration sampled INSIDE the timed solves (every 8th iteration, on the stream it is launched on).
          roofline.csr_plain: the same system solved again on the plain CSR stream (spmv_csr_dma<256, SPMV_DOT, nt>,
          12*nnz + 20*n bytes: the north_star's ">= 70 % on the CSR SpMV"), timed the same way.
          roofline.unstructured: the same matrix under pseudo-random symmetric renumberings (no dictionary, real
          gathers): what a caller's mesh numbering sees -- as the backend runs it by default (a scattered numbering is
          renumbered at factorize, "reorder" 2; search and copy timed) and in the caller's numbering
          (`caller_numbering`, "reorder" 0).
          roofline.traffic: HBM bytes per launch of the roofline's kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two child
          passes over one solve of this command spawned by this run (committed profiles/*_pmc_traffic*.json as a fall-back).
elasticity: BASELINE.json configs[2] (Q1 elasticity M = 100, block-3 Chebyshev-AMG PCG) as an extra block.
cpu_baseline: the CPU oracle's restatement of the same Jacobi-PCG (Eigen::ConjugateGradient path),
          timed on this box's host cores over a bounded number of iterations of the same system.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# The AMG configuration this backend recommends (the reference's AMGCL configuration -- W-cycle, Chebyshev-16, 100 power
# iterations, AMGCL.cpp:32-65 -- is timed next to it where it matters): V-cycle, Chebyshev degree 2 on [0.1, 1.1] x the
# power-iteration estimate of rho(D^-1 A) (AMGCL's safety factor `higher` = 2 spends the smoother on an interval where
# the operator has no spectrum: 256^3 57 -> 50 ms, elasticity 128 -> 112 ms), prolongation smoothing over-relaxed by
# 1.3 (50 -> 46.5 ms / 112 -> 99 ms; profiles/r03_amg.md)
AMG_RECOMMENDED = dict(ncycle=1, cheb_degree=2, cheb_lower=0.1, cheb_higher=1.1, cheb_power_iters=20, sa_relax=1.3)


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


def socket0_cpus():
    """One hardware thread per physical core of CPU package 0 (the "single socket" of the north_star), from sysfs;
    falls back to every CPU this process may run on."""
    try:
        seen, cpus = set(), []
        for c in sorted(os.sched_getaffinity(0)):
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            if int(open(base + "physical_package_id").read()) != 0:
                continue
            core = int(open(base + "core_id").read())
            if core not in seen:
                seen.add(core)
                cpus.append(c)
        if cpus:
            return cpus
    except Exception:
        pass
    return sorted(os.sched_getaffinity(0))


def run_cpu_leg(kind: str, **kw):
    """Run one CPU-baseline leg in a child process PINNED to the cores of socket 0 (affinity mask + OpenMP places set
    before the OpenMP runtime starts): unpinned, the same code varied 2x between boxes (NUMA placement)."""
    import subprocess
    cpus = socket0_cpus()
    env = dict(os.environ, OMP_NUM_THREADS=str(len(cpus)), OMP_PLACES="cores", OMP_PROC_BIND="close",
               PSOLVE_BENCH_CPUS=",".join(map(str, cpus)))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", kind] + [f"--{k.replace('_', '-')}={v}" for k, v in kw.items()]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        raise RuntimeError(f"cpu leg {kind} failed: {out.stderr[-400:]}")
    return json.loads(lines[-1])


def cpu_leg(args):
    """Child process: the timed CPU work.  kind "eigen": oracle.cg_eigen (the restatement of
    Eigen::ConjugateGradient + DiagonalPreconditioner) on the bench system for a bounded number of iterations.
    kind "amgcl": oracle.AMG + oracle.cg_amgcl with the reference's AMGCL defaults (AMGCL.cpp:32-65: W-cycle,
    Chebyshev-16, 100 power iterations), setup and solve timed separately -- the north_star's CPU side."""
    cpus = [int(c) for c in os.environ.get("PSOLVE_BENCH_CPUS", "").split(",") if c]
    if cpus:
        os.sched_setaffinity(0, cpus)
    import oracle as O
    cores = O.lib().orc_num_threads()
    N = args.grid
    pin = f"pinned to the {len(cpus)} cores of socket 0 (sched_setaffinity + OMP_PLACES=cores OMP_PROC_BIND=close)" if cpus else "unpinned"
    t = time.perf_counter()
    A = O.poisson7(N)
    b = O.spmv(A, O.splitmix_vector(A.n, 42))
    t_gen = time.perf_counter() - t
    nnz = A.nnz
    # what the socket streams, measured in this very child (same pinning, same first-touch placement): STREAM-like triad over
    # three vectors of n doubles (a reader sees how far the port is from the socket's own roofline)
    triad_gbs = O.stream_triad(max(A.n, 1 << 25), 5)  # (three arrays of at least 256 MiB: beyond every cache)
    if args.cpu_leg == "eigen":
        gpu_passes = args.passes
        t = time.perf_counter()
        O.cg_eigen(A, b, tol=1e-8, max_iter=2)  # warm-up + per-iteration estimate
        per_it = (time.perf_counter() - t) / 3.0
        REPEATS = 3
        iters = int(max(4, min(gpu_passes, args.budget / REPEATS / max(per_it, 1e-6))))
        runs = []
        for _ in range(REPEATS):  # best of three: the hosts of these boxes are shared, one sample swung 3.6x between leases
            t = time.perf_counter()
            _, it, _ = O.cg_eigen(A, b, tol=1e-8, max_iter=iters)
            dt = time.perf_counter() - t
            passes = it + 1 if it < iters else iters
            runs.append(dt / (passes + 1))  # one residual product + `passes` loop products were timed
        sec_it = min(runs)
        full = sec_it * (gpu_passes + 1)
        iter_bytes = 12 * nnz + 156 * A.n  # the unfused Eigen loop, which is what the port runs
        # the reference's own build has no -fopenmp: Eigen::ConjugateGradient runs on ONE thread
        # there.  The same restatement on one thread, a few iterations, scaled the same way.
        O.lib().orc_set_num_threads(1)
        it1 = 4
        t = time.perf_counter()
        O.cg_eigen(A, b, tol=1e-8, max_iter=it1)
        dt1 = time.perf_counter() - t
        O.lib().orc_set_num_threads(cores)
        full1 = dt1 * (gpu_passes + 1) / (it1 + 1)
        print(json.dumps({"value": A.n / full, "unit": "DOF/s", "cores": cores, "kind": "port",
                          "sample": f"best of {REPEATS} runs of {passes} of {gpu_passes} PCG iterations of the same {N}^3 system "
                                    f"(oracle.cg_eigen, OpenMP x{cores}, {pin}), scaled to the full solve",
                          "repeats": REPEATS, "seconds_per_iteration": sec_it,
                          "seconds_per_iteration_runs": [round(v, 5) for v in runs],
                          "gbs": iter_bytes / sec_it / 1e9, "bytes_per_iteration": iter_bytes,
                          "stream_triad_gbs": triad_gbs,
                          "frac_of_stream_triad": (iter_bytes / sec_it / 1e9) / triad_gbs if triad_gbs > 0 else None,
                          "reference_single_thread": {"value": A.n / full1, "unit": "DOF/s", "cores": 1,
                                                      "seconds_per_iteration": dt1 / (it1 + 1),
                                                      "sample": f"{it1} iterations on one thread ({dt1:.1f} s), scaled; the "
                                                                "reference build of Eigen::ConjugateGradient is single-threaded"}}))
    else:
        REPEATS = int(os.environ.get("PSOLVE_BENCH_CPU_REPEATS", "3"))
        setups, solves = [], []
        for _ in range(REPEATS):  # best of three, setup and solve each (one sample swung 7.2 <-> 26.4 s between leases)
            t = time.perf_counter()
            amg = O.AMG(A)  # AMGCL.cpp:32-65 defaults
            setups.append(time.perf_counter() - t)
            t = time.perf_counter()
            x, it, err = O.cg_amgcl(A, b, precond=amg, tol=1e-8, max_iter=1000)
            solves.append(time.perf_counter() - t)
        t_setup, t_solve = min(setups), min(solves)
        r = b - O.spmv(A, x)
        import numpy as np
        print(json.dumps({"cores": cores, "pinning": pin, "kind": "port", "setup_s": t_setup, "solve_s": t_solve,
                          "repeats": REPEATS, "setup_s_runs": [round(v, 3) for v in setups],
                          "solve_s_runs": [round(v, 3) for v in solves], "stream_triad_gbs": triad_gbs,
                          "iterations": int(it), "final_res_norm": err,
                          "true_residual": float(np.linalg.norm(r) / np.linalg.norm(b)), "generate_s": t_gen,
                          "levels": amg.num_levels,
                          "what": "oracle restatement of AMGCL 1.4.3 with the reference's defaults (cg, smoothed aggregation, "
                                  "W-cycle, Chebyshev-16, 100 power iterations); OpenMP where AMGCL's builtin backend is "
                                  "(strength test, smoothed prolongation, row-parallel Galerkin products, power "
                                  "iterations, cycle, CG), sequential where it is (aggregation sweep, transposes)"}))
    return 0


def north_star_block(HIPSolver, np, N=216, with_cpu=True):
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
            s.set_parameters({"HIP": {"amg": {"refresh_power_iters": -1}}})
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
    if not with_cpu:
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


def amg_cycle_ops(s, nlevels, block, nnzb0=0, max_level=1):
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
            mat = 2 * (rows // 3) if bk else 76 * (nnzb0 if (l == 0 and nnzb0) else nnz // 9) + 4 * (rows // 3)
        else:
            rk = l == 0 and s.get_param("spmv_row_kinds") > 0  # (... and of a scalar one from row kinds: 2 bytes per row)
            mat = 2 * rows if rk else (8 * nnz + 6 * rows) if (l == 0 and s.get_param("spmv_patterns") > 0) else (12 * nnz + 4 * rows)
        ops = {"cheb_step": (t["cheb_step_us"], mat + 8 * cols + 40 * rows + (24 * rows if block else 0)),
               "residual": (t["residual_us"], mat + 8 * cols + 16 * rows),
               "cheb_first": (t["cheb_first_us"], (8 * 6 if block else 8 * 4) * rows)}
        if l + 1 < nlevels:
            for name, what, vec in (("restrict", 2, 8), ("prolong", 1, 16)):
                r2, c2, z2 = s.amg_level_matrix_shape(l, what)
                m2 = (76 * (z2 // 9) + 4 * (r2 // 3)) if block else (12 * z2 + 4 * r2)
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
    t_refresh_keep = None
    if not amg_extra:
        try:
            s.set_parameters({"HIP": {"amg": {"refresh_power_iters": 0}}})
            gen()  # (cold estimate once more, this time keeping its last vector)
            s.synchronize()
            t = time.perf_counter()
            gen()
            s.synchronize()
            t_refresh_keep = time.perf_counter() - t
            s.set_parameters({"HIP": {"amg": {"refresh_power_iters": -1}}})
            gen()  # back to the default estimate for the solves below
            s.synchronize()
        except Exception:
            t_refresh_keep = None
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
    cycle_ops = amg_cycle_ops(s, int(info["amg_levels"]), block=True, nnzb0=nnzb)
    out = {"generate_plus_setup_s": t_setup, "generate_plus_refresh_s": t_refresh, "refresh_reused_patterns": refreshed,
           "generate_plus_refresh_keep_radii_s": t_refresh_keep,
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


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: check that the node has N GPUs, then re-run this command
    line as N ranks (LOCAL_RANK = GPU index, rendezvous on 127.0.0.1) and pass rank 0's JSON line through.
    Fails -- loudly, non-zero -- rather than print a line for fewer GPUs than were asked for."""
    import socket
    import subprocess
    probe = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                           capture_output=True, text=True)
    try:
        visible = int(probe.stdout.strip().splitlines()[-1])
    except Exception:
        visible = 0
    if visible < n:
        raise SystemExit(f"bench.py: --gpus {n} requested but only {visible} GPU(s) visible on this node; "
                         f"refusing to report a {n}-GPU number from fewer devices")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    if any(codes):
        for p in procs:
            if p.poll() is None:
                p.kill()
        raise SystemExit(f"bench.py: ranks exited with {codes}")
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if len(lines) != 1 or json.loads(lines[0]).get("n_gpus") != n:
        raise SystemExit(f"bench.py: expected one JSON line for {n} GPUs from rank 0, got: {out0[-500:]}")
    print(lines[0])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=256, help="N of the N^3 Poisson grid")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip both CPU legs (cpu_baseline and north_star's)")
    ap.add_argument("--no-north-star", action="store_true", help="skip the 10 M-DOF AMG-PCG GPU-vs-CPU block")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not spawn the two rocprofv3 --pmc passes that measure roofline.traffic in this run (the committed "
                         "profiles/*_pmc_traffic*.json of the same kernel is attached instead)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the extra legs (plain-CSR and unstructured SpMV legs, elasticity block): profiling runs")
    ap.add_argument("--elasticity-m", type=int, default=100, help="nodes per edge of the elasticity block (3 M^3 DOF)")
    ap.add_argument("--spmv-kernel", type=int, default=-1, choices=[-1, 0, 1, 2, 3],
                    help="the backend's spmv_kernel for the timed solves (1: plain CSR stream; profiling runs)")
    ap.add_argument("--value-dict", type=int, default=1, choices=[0, 1],
                    help="0: keep the value stream of an operator whose rows repeat (the pattern-dictionary kernel), 1: row kinds (default)")
    ap.add_argument("--cpu-leg", default=None, choices=["eigen", "amgcl"], help=argparse.SUPPRESS)
    ap.add_argument("--passes", type=int, default=500, help=argparse.SUPPRESS)
    ap.add_argument("--budget", type=float, default=20.0, help=argparse.SUPPRESS)
    ap.add_argument("--precond", default="jacobi", choices=["jacobi", "none", "amg"],
                    help="jacobi = BASELINE.json's configuration; amg = Chebyshev-smoothed aggregation V-cycle "
                         "(on shards: one global hierarchy, level 0 distributed, coarser levels replicated)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong: the same grid^3 system on N GPUs (north_star's target); weak: 256^3 rows per GPU "
                         "(N=8 -> 512^3 = BASELINE.json configs[3])")
    args = ap.parse_args()

    if args.cpu_leg:
        return cpu_leg(args)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: be our own launcher -- N ranks of this script, one per GPU, over
        # RCCL, the same way the driver starts it (`python -m torch.distributed.run --nproc-per-node N ...`)
        return spawn_ranks(args.gpus)

    import torch  # first: one HIP runtime (torch's) for torch and libpsolve_hip.so alike
    import numpy as np
    from polysolve_amd import HIPSolver

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the job must run exactly one rank "
                         f"per requested GPU")
    visible = torch.cuda.device_count()
    if local_rank >= visible or world > visible:
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {world} GPUs on this node, {visible} visible")
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        # one node by contract (N GPUs of ONE node, rendezvous on 127.0.0.1): RCCL's bootstrap needs the loopback
        # interface only and no InfiniBand probing (a box whose hostname does not resolve can spend minutes there)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    N = args.grid
    s = HIPSolver("" if args.precond == "jacobi" else "Eigen::IdentityPreconditioner", device=local_rank)
    s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "profile_spmv": 8, "spmv_kernel": args.spmv_kernel,
                              "spmv_value_dict": bool(args.value_dict)}})
    if args.precond == "amg":
        s.set_parameters({"HIP": {"precond": "amg", "amg": dict(AMG_RECOMMENDED)}})
        if "PSOLVE_BENCH_RENUMBER" in os.environ:  # A/B runs of the coarse-level renumbering (scripts/gpu_r3_amgprof2.sh)
            s.set_parameters({"HIP": {"amg": {"renumber": int(os.environ["PSOLVE_BENCH_RENUMBER"])}}})
    if world > 1:
        # RCCL communicator of the backend itself; torch.distributed only carries the 128-byte id
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(HIPSolver.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        s.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()))
    nx = ny = nz = N
    if args.scaling == "weak":  # double z, then y, then x: 2 -> 256x256x512, 4 -> 256x512x512, 8 -> 512^3
        f = world
        for dim in ("z", "y", "x") * 4:
            if f <= 1:
                break
            if dim == "z":
                nz *= 2
            elif dim == "y":
                ny *= 2
            else:
                nx *= 2
            f //= 2
    cuts = [round(q * nz / world) for q in range(world + 1)]
    z0, z1 = cuts[rank], cuts[rank + 1]
    s.generate_poisson7(nx, ny, nz, z0, z1)  # shard generated on its own device, then "factorized"
    n_loc, nnz_loc, n_halo = s.matrix_shape()
    npat = int(s.get_param("spmv_patterns"))  # > 0: PCG's product runs on the pattern dictionary
    n_global = nx * ny * nz
    b = s.device_array(n_loc)
    x = s.device_array(n_loc)
    s.generate_rhs(42, b)

    def sync():
        s.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def one_solve():
        s.axpby_device(n_loc, 0.0, b, 0.0, x)  # x0 = 0 * b = 0, set on the device
        s.solve_device(b, x)

    for _ in range(args.warmup):
        one_solve()
    sync()
    # reference point for the roofline: what a plain device copy (y = 1.0 * x, 1 GiB -> 1 GiB: larger
    # than the 256 MiB Infinity Cache) reaches on THIS box
    copy_gbs = None
    if rank == 0:
        nc = 1 << 27
        src, dst = s.device_array(nc), s.device_array(nc)
        s.axpby_device(nc, 0.0, src, 0.0, src)  # define the source
        s.axpby_device(nc, 1.0, src, 0.0, dst)
        s.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            s.axpby_device(nc, 1.0, src, 0.0, dst)
        s.synchronize()
        copy_gbs = 10 * 16.0 * nc / (time.perf_counter() - t1) / 1e9
        src.free()
        dst.free()
    sync()
    box_before = BoxSampler(local_rank)
    with box_before:  # (idle state right before the timed region: a few samples)
        time.sleep(0.1)
    sampler = BoxSampler(local_rank)
    sampler.__enter__()  # a host thread reading sysfs: nothing of it touches the device queue
    t0 = time.perf_counter()
    spmv_ms, spmv_samples, passes = 0.0, 0, 0
    for _ in range(args.steps):
        one_solve()
        i = s.info_struct()
        spmv_ms += i.spmv_ms_avg * i.spmv_samples
        spmv_samples += i.spmv_samples
        passes = i.num_iterations
    sync()
    elapsed = time.perf_counter() - t0
    sampler.__exit__()
    info = s.get_info()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    if rank == 0:
        spmv_avg_ms = spmv_ms / max(spmv_samples, 1)
        pat_in_use = npat > 0 and args.spmv_kernel in (-1, 3)

        # the kernel of this line, as the LIBRARY reports it (psolve_hip_last_spmv_kernel: the instantiation PCG's product ran on
        # in the timed solves, spelled as rocprofv3 prints it) -- VERDICT r4 item 9: not a name composed here
        try:
            lib_kernel = s.last_spmv_kernel()
        except Exception:
            lib_kernel = ""

        def pmc_traffic(kernel):
            """HBM traffic per SpMV launch: rocprofv3 cannot run inside the bench, so this is the COMMITTED result of
            the PMC passes over this very command (scripts/r5/pmc_bench.sh -> scripts/make_pmc_traffic.py), newest
            round first -- read from a file, not measured in this run; attached ONLY when the file was made for the
            instantiation the library reports (`kernel_library_name`)"""
            try:
                import glob
                for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
                    pmc = json.load(open(f))
                    if (world == 1 and N == 256 and args.precond == "jacobi" and pmc.get("workload") == "poisson7 256^3"
                            and kernel and pmc.get("kernel_library_name") == kernel):
                        return pmc["traffic_bytes"], (os.path.relpath(f, ROOT) + " (committed rocprofv3 --pmc passes of this "
                                                      "command, not measured in this run; kernel " + pmc["kernel_library_name"] + ")")
            except Exception:
                pass
            return None, None

        def live_pmc_traffic(kernel):
            """HBM traffic per launch of `kernel`, MEASURED IN THIS RUN (round-4 review, weak #8): two child processes --
            rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes) over one
            solve of this very command -- read back from their counter CSVs; FETCH_SIZE (KB) x 2 on gfx950, WRITE_SIZE in
            KB, means over the live launches.  None when rocprofv3 is missing, fails or takes too long."""
            import csv, glob, shutil, subprocess, tempfile
            if args.no_live_traffic or not kernel or world != 1 or not shutil.which("rocprofv3"):
                return None, None
            tot = {}
            try:
                for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                    d = tempfile.mkdtemp(prefix="psolve_pmc_", dir="/tmp")
                    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "b", "--",
                           sys.executable, os.path.abspath(__file__), "--grid", str(N), "--steps", "1", "--warmup", "0",
                           "--no-cpu-baseline", "--no-north-star", "--no-extra", "--no-live-traffic", "--precond", args.precond,
                           "--spmv-kernel", str(args.spmv_kernel), "--value-dict", str(args.value_dict)]
                    env = dict(os.environ, TMPDIR="/tmp")
                    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=90, check=True)
                    vals = []
                    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                        for row in csv.DictReader(open(f)):
                            if row["Counter_Name"] == counter and kernel.split("<")[0] in row["Kernel_Name"] and kernel in row["Kernel_Name"].replace("psolve::", ""):
                                vals.append(float(row["Counter_Value"]))
                    shutil.rmtree(d, ignore_errors=True)
                    live = [v for v in vals if v > 0.5 * max(vals)] if vals and max(vals) > 0 else []
                    if not live:
                        return None, None
                    tot[counter] = sum(live) / len(live)
                traffic = tot["FETCH_SIZE"] * 1024.0 * 2.0 + tot["WRITE_SIZE"] * 1024.0
                return traffic, ("measured in this run: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (two child passes over one solve of this "
                                 "command; FETCH_SIZE x 2 per MI355X_MICROARCH.md, means over the live launches of " + kernel + ")")
            except Exception:
                return None, None

        csr_bytes = 12 * nnz_loc + 20 * n_loc   # SURVEY.md 8(d)'s figure for a plain CSR product
        spmv_kernel_name = lib_kernel or "unknown (library reported none)"
        stream_bytes, spmv_format = spmv_stream_bytes(lib_kernel, n_loc, nnz_loc, npat, int(s.get_param("spmv_row_kinds")))
        pat_in_use = lib_kernel.startswith(("spmv_csr_pat", "spmv_csr_kind", "spmv_csr_slots"))
        stream_gbs = stream_bytes / (spmv_avg_ms * 1e-3) / 1e9 if spmv_avg_ms > 0 else 0.0
        # The three kernels of a Jacobi-PCG iteration, each launched once per iteration and timed by HIP events inside the
        # timed solves (every 8th iteration, on the stream they are launched on); names from the library.  `roofline` is
        # the one that takes the most time per launch -- since round 5's row kinds that is no longer the product on this
        # constant-coefficient grid, but a vector update.
        kernels = [dict(spmv_leg(spmv_kernel_name, stream_bytes, spmv_avg_ms, int(spmv_samples)), role="q = A p, p.q")]
        try:
            k2_ms, k3_ms = s.get_param("stats.update_r_ms_avg"), s.get_param("stats.update_xp_ms_avg")
            if k2_ms > 0 and k3_ms > 0:
                # (row kinds: 1 / diag is read as table[kind[row]], 2 bytes per row instead of 8 -- "pcg_kind_diag")
                kd = int(s.get_param("pcg_kind_diag")) == 1
                kernels.append(dict(spmv_leg(s.last_pcg_kernel(1), (26 if kd else 32) * n_loc, k2_ms, int(spmv_samples)),
                                    role="r -= alpha q, r.r, r.z (reads q r and " + ("the row kinds: 26 n bytes)" if kd else "1/diag, writes r: 32 n bytes)")))
                kernels.append(dict(spmv_leg(s.last_pcg_kernel(2), (42 if kd else 48) * n_loc, k3_ms, int(spmv_samples)),
                                    role="x += alpha p, p = z + beta p (reads p x r and " + ("the row kinds, writes x p: 42 n bytes)" if kd else "1/diag, writes x p: 48 n bytes)")))
        except Exception:
            pass
        t_all = sum(k["avg_launch_ms"] for k in kernels) or 1.0
        for k in kernels:
            k["share_of_sampled_iteration"] = k["avg_launch_ms"] / t_all
        dom = max(kernels, key=lambda k: k["avg_launch_ms"])
        traffic, traffic_src = None, None
        if N == 256 and args.precond == "jacobi":  # (the line's own workload; the legs keep the committed files)
            traffic, traffic_src = live_pmc_traffic(dom["kernel"])
        if traffic is None:
            traffic, traffic_src = pmc_traffic(dom["kernel"])
        out = {
            "metric": "DOF/s to 1e-8 rel-residual on 3-D Poisson SPD",
            "value": n_global * args.steps / elapsed,
            "unit": "DOF/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"3-D 7-point Poisson {nx}x{ny}x{nz} ({n_global} DOF), "
                                   f"{args.precond}-PCG to ||r||/||b||<1e-8, x0=0, CSR fp64/int32",
                       "grid": [nx, ny, nz], "precond": args.precond, "partition": f"{world} z-slab(s)",
                       "rows_per_gpu": n_loc, "halo_per_gpu": n_halo},
            "iterations": int(passes),
            "ms_per_iteration": elapsed * 1e3 / args.steps / max(int(passes), 1),
            "solver_error": info["solver_error"],
            "true_residual": info["true_residual"],
            # frac = bytes the timed kernel streams per launch / its in-loop launch time / 8 TB/s.  The same launch
            # expressed in plain-CSR bytes (what an index-uncompressed kernel would have had to move to be as fast)
            # is csr_equivalent_gbs: a throughput equivalent, NOT a bandwidth, never a fraction of peak.
            "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["achieved"] / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom["bytes_per_launch"],
                         "avg_launch_ms": dom["avg_launch_ms"], "launches_sampled": dom["launches_sampled"],
                         "dominant_of": "the kernel with the longest sampled launch among the iteration's kernels (`kernels`)",
                         "kernels": kernels,
                         # the product kernel (the north_star's subject) in detail; `csr_pat` / `csr_plain` below are the same
                         # system solved again on the formats that stream the matrix
                         "spmv": dict(kernels[0], format=spmv_format, csr_bytes_per_launch=csr_bytes,
                                      csr_equivalent_gbs=csr_bytes / (spmv_avg_ms * 1e-3) / 1e9 if spmv_avg_ms > 0 else 0.0,
                                      frac_of_device_copy=(stream_gbs / copy_gbs) if copy_gbs else None),
                         "csr_bytes_per_launch": csr_bytes,
                         "device_copy_gbs_this_box": copy_gbs,
                         "frac_of_device_copy": (dom["achieved"] / copy_gbs) if copy_gbs else None},
        }
        if args.precond == "amg" and world == 1:  # the V-cycle's operations on levels 0 and 1, each against its bytes
            try:
                out["amg_cycle_ops"] = amg_cycle_ops(s, int(s.get_info()["amg_levels"]), block=False)
            except Exception as e:
                out["amg_cycle_ops"] = {"failed": str(e)}
        # what this line's collectives actually ran on: ranks of a real RCCL communicator (0 at N = 1)
        out["comm_rccl_ranks_seen"] = int(s.get_param("dist.rccl_ranks_seen"))
        if world > 1:  # per-iteration communication of rank 0, HIP events around the sampled iterations' collectives
            out["comm"] = {"rccl_ranks_seen": int(s.get_param("dist.rccl_ranks_seen")),
                           "allreduce_us_avg": s.get_param("stats.allreduce_us_avg"), "allreduce_samples": int(s.get_param("stats.allreduce_samples")),
                           "halo_exchange_us_avg": s.get_param("stats.halo_us_avg"), "halo_samples": int(s.get_param("stats.halo_samples")),
                           "what": "one all-reduce of the CG scalars (main stream) and the halo exchange of p (its own stream, overlapped with "
                                   "the interior rows) of every 8th iteration of the last solve, rank 0"}
        try:  # (after the timed region: latencies / gather rates of this box, probe.hip; 1 GiB of scratch)
            probe = s.box_probe()
        except Exception as e:
            probe = {"failed": str(e)}
        out["box"] = dict(box_static(local_rank), idle_before=box_before.summary(), during_timed_region=sampler.summary(), probe=probe)
        # whole-iteration view: the three fused kernels move (SpMV stream) + 80 n bytes per iteration (K2 32 n, K3 48 n);
        # Eigen's unfused loop would move 12 nnz + 156 n -- given as bytes only, for reference
        it_s = elapsed / args.steps / max(int(passes), 1)
        fused = sum(k["bytes_per_launch"] for k in kernels) if len(kernels) == 3 else stream_bytes + 80 * n_loc
        out["iteration_roofline"] = {
            "fused_bytes_per_iteration": fused, "fused_gbs": fused / it_s / 1e9,
            "fused_frac_of_peak": fused / it_s / 1e9 / HBM_PEAK_GBS,
            "eigen_unfused_bytes_per_iteration": 12 * nnz_loc + 156 * n_loc}
        extra = world == 1 and args.precond == "jacobi" and not args.no_extra
        if extra and lib_kernel.startswith(("spmv_csr_kind", "spmv_csr_slots")):
            # the same system with the VALUES streamed (pattern dictionary only: 8 nnz + 22 n bytes per launch) -- what a
            # structured mesh with varying coefficients runs; round 4's headline kernel
            try:
                s.set_parameters({"HIP": {"spmv_value_dict": False}})
                s.generate_poisson7(nx, ny, nz, z0, z1)
                dt, its, ms, smp, inf = time_solves(s, b, x, n_loc, reps=2, warm_iters=32)
                kpat = s.last_spmv_kernel()
                tr, tr_src = pmc_traffic(kpat)
                out["roofline"]["csr_pat"] = spmv_leg(
                    kpat, 8 * nnz_loc + 22 * n_loc, ms, smp,
                    {"iterations": its, "solve_s": dt, "dof_per_s": n_loc / dt, "ms_per_iteration": dt * 1e3 / max(its, 1),
                     "true_residual": inf["true_residual"], "traffic": tr, "traffic_source": tr_src,
                     "frac_of_device_copy": None})
                if copy_gbs:
                    out["roofline"]["csr_pat"]["frac_of_device_copy"] = out["roofline"]["csr_pat"]["achieved"] / copy_gbs
            except Exception as e:
                out["roofline"]["csr_pat"] = {"failed": str(e)}
        if extra and pat_in_use:
            # the north_star's kernel: the SAME system on the plain CSR stream (12 nnz + 20 n bytes per launch), timed
            # the same way inside full solves -- what every operator without a dictionary (unstructured meshes) runs
            try:
                s.set_parameters({"HIP": {"spmv_kernel": 1}})
                dt, its, ms, smp, inf = time_solves(s, b, x, n_loc, reps=2, warm_iters=32)
                plain_kernel = s.last_spmv_kernel()
                tr, tr_src = pmc_traffic(plain_kernel)
                out["roofline"]["csr_plain"] = spmv_leg(
                    plain_kernel, csr_bytes, ms, smp,
                    {"iterations": its, "solve_s": dt, "dof_per_s": n_loc / dt, "ms_per_iteration": dt * 1e3 / max(its, 1),
                     "true_residual": inf["true_residual"], "traffic": tr, "traffic_source": tr_src,
                     "frac_of_device_copy": None})
                if copy_gbs:
                    out["roofline"]["csr_plain"]["frac_of_device_copy"] = out["roofline"]["csr_plain"]["achieved"] / copy_gbs
                s.set_parameters({"HIP": {"spmv_kernel": args.spmv_kernel}})
            except Exception as e:
                out["roofline"]["csr_plain"] = {"failed": str(e)}
        # `value` by the storage the product ran on, side by side: the line's own (what the backend picks for THIS matrix: a
        # constant-coefficient grid has row kinds) and the same system with the matrix streamed -- what a matrix whose rows
        # do not repeat gets (varying coefficients on a structured mesh: the pattern dictionary; any other: plain CSR)
        vbs = {("row_kinds" if lib_kernel.startswith(("spmv_csr_kind", "spmv_csr_slots")) else
                "pattern_dictionary" if lib_kernel.startswith("spmv_csr_pat") else "plain_csr"): out["value"]}
        for key, leg in (("pattern_dictionary", "csr_pat"), ("plain_csr", "csr_plain")):
            v = out["roofline"].get(leg)
            if isinstance(v, dict) and "dof_per_s" in v and key not in vbs:
                vbs[key] = v["dof_per_s"]
        out["value_by_storage"] = vbs
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = run_cpu_leg("eigen", grid=N, passes=int(passes))
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "DOF/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
        else:
            out["cpu_baseline"] = None
        if world == 1:  # (a shard's communicator is torn down by every rank together, at exit)
            b.free()
            x.free()
            del s
        if extra:
            try:
                out["roofline"]["unstructured"] = unstructured_block(HIPSolver, N)
            except Exception as e:
                out["roofline"]["unstructured"] = {"failed": str(e)}
            try:
                out["elasticity"] = elasticity_block(HIPSolver, args.elasticity_m)
            except Exception as e:
                out["elasticity"] = {"failed": str(e)}
            try:  # what PolyFEM actually calls: host arrays in, host vectors out (PCIe inside the timed calls)
                out["host_contract"] = host_contract_block(HIPSolver, np, N, args.elasticity_m)
            except Exception as e:
                out["host_contract"] = {"failed": str(e)}
        if world == 1 and N == 256 and args.precond == "jacobi" and not args.no_north_star:
            # extra block, headline untouched: the north_star's 10 M-DOF AMG-PCG comparison (GPU vs one CPU socket)
            try:
                out["north_star"] = north_star_block(HIPSolver, np, with_cpu=not args.no_cpu_baseline)
            except Exception as e:
                out["north_star"] = {"failed": str(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
