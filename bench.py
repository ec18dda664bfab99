"""bench.py -- the headline benchmark of BASELINE.json on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid 256] [--storage csr|auto]

metric  : DOF/s to 1e-8 relative residual on 3-D 7-point Poisson (SPD), Jacobi-PCG
workload: BASELINE.json configs[1] -- N=256^3 (16.8 M DOF) on one MI355X; with --gpus N the SAME system is
          row-partitioned (z-slabs) over N ranks, one process per GPU ("scaling": "strong", the north_star's target).
storage : "csr" (default): the matrix is streamed as plain CSR (fp64 values, int32 columns) by spmv_csr_dma -- the
          north_star's "CSR double SPD matrices", what ANY caller's matrix gets.  "auto": what the backend picks for
          THIS matrix (a constant-coefficient grid collapses to row kinds and streams no matrix -- a special case,
          reported in `value_by_storage`, never the headline).
step    : one full solve (x0 = 0 -> ||r||/||b|| < 1e-8) with matrix, b and x resident in HBM.
value   : n_global * steps / wall time of the K timed solves (max over ranks).
roofline: the dominant kernel of the timed solves -- PCG's CSR SpMV, SURVEY.md 8(d): 12 nnz + 20 n bytes per launch -- over
          its HIP-event duration sampled INSIDE the timed solves: every 8th iteration the product is launched with a pair of
          events of its own (hipExtLaunchKernelGGL: the kernel's begin and end timestamps on the stream it runs on -- the
          figure rocprofv3's kernel trace reports; events recorded around the launch also time the dispatch gap).
          traffic: HBM bytes per launch from the committed rocprofv3 --pmc passes over this command
          (profiles/r*_pmc_traffic*.json, attached only when made for the kernel instantiation the library reports);
          --live-traffic measures it in this run instead (two rocprofv3 child passes).
cpu_baseline: the CPU oracle's restatement of the same Jacobi-PCG (Eigen::ConjugateGradient path) on this box's host
          cores (socket 0), over a bounded number of iterations of the same system; `tuned_value`: the same recurrence as
          a tuned CPU code would run it (fused passes, first-touch placement).

The LAST stdout line is the bench line: flat, scalars only inside `roofline` / `cpu_baseline`, a few KB.  Everything
else (the other storages, unstructured renumberings, configs[2] elasticity, the host contract, the north_star's 10 M-DOF
AMG comparison, the state of the box) goes to bench_detail.json (and gpurun_out/bench_detail.json where that directory
exists), with a short "# detail" digest on stderr.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


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


def cpu_quota():
    """CPUs' worth of time the container's cgroup grants (cpu.max: "<quota> <period>" | "max ..."); None = unlimited.
    The GPU boxes of this pool show 256 hardware threads and a quota of 16: more runnable threads than that are
    throttled (round 6: 64 threads ran the same loop 2.6x slower than 16 spread over the socket)."""
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else float(t.split()[0]) / float(t.split()[1])),):
        try:
            return parse(open(path).read())
        except Exception:
            pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        return None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
    except Exception:
        return None


def run_cpu_leg(kind: str, **kw):
    """Run one CPU-baseline leg in a child process on socket 0 (the "single socket" of the north_star): affinity = one hardware
    thread per physical core of package 0, threads = min(those cores, the cgroup's CPU quota), spread over the socket's
    places (OMP_PROC_BIND=spread: 16 threads packed on two CCDs reach half the bandwidth of 16 spread over eight), all set
    before the OpenMP runtime starts."""
    import subprocess
    cpus = socket0_cpus()
    quota = cpu_quota()
    threads = max(1, min(len(cpus), int(quota))) if quota else len(cpus)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PLACES="cores", OMP_PROC_BIND="spread",
               PSOLVE_BENCH_CPUS=",".join(map(str, cpus)), PSOLVE_BENCH_CPU_QUOTA=str(quota if quota else 0))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", kind] + [f"--{k.replace('_', '-')}={v}" for k, v in kw.items()]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        raise RuntimeError(f"cpu leg {kind} failed: {out.stderr[-400:]}")
    return json.loads(lines[-1])


def cpu_leg(args):
    """Child process: the timed CPU work (the ONLY place bench.py touches oracle/).
    kind "eigen": oracle.cg_eigen (the restatement of Eigen::ConjugateGradient + DiagonalPreconditioner: unfused, 12 nnz +
    156 n bytes per iteration) and oracle.cg_jacobi_tuned (the same recurrence as tuned CPU code runs it: three fused
    passes, 12 nnz + 100 n bytes, arrays first-touched by the threads that stream them) on the bench system for a bounded
    number of iterations.  kind "amgcl": oracle.AMG + oracle.cg_amgcl with the reference's AMGCL defaults
    (AMGCL.cpp:32-65), setup and solve timed separately -- the north_star's CPU side."""
    cpus = [int(c) for c in os.environ.get("PSOLVE_BENCH_CPUS", "").split(",") if c]
    if cpus:
        os.sched_setaffinity(0, cpus)
    import oracle as O
    cores = O.lib().orc_num_threads()
    N = args.grid
    quota = float(os.environ.get("PSOLVE_BENCH_CPU_QUOTA", "0") or 0)
    pin = (f"{cores} threads spread over the {len(cpus)} cores of socket 0" if cpus else "unpinned") + (f" (cgroup CPU quota {quota:g})" if quota else "")
    t = time.perf_counter()
    A = O.poisson7(N)
    b = O.spmv(A, O.splitmix_vector(A.n, 42))
    t_gen = time.perf_counter() - t
    nnz = A.nnz
    # what the socket streams, measured in this very child (same pinning, same first-touch placement)
    triad_gbs = O.stream_triad(max(A.n, 1 << 25), 5)  # (three arrays of at least 256 MiB: beyond every cache)
    if args.cpu_leg == "eigen":
        gpu_passes = args.passes
        REPEATS = 3

        def timed(fn, budget):
            t = time.perf_counter()
            fn(2)  # warm-up + per-iteration estimate
            per_it = (time.perf_counter() - t) / 3.0
            iters = int(max(4, min(gpu_passes, budget / REPEATS / max(per_it, 1e-6))))
            runs, passes = [], iters
            for _ in range(REPEATS):  # best of three: the hosts of these boxes are shared
                t = time.perf_counter()
                it = fn(iters)
                dt = time.perf_counter() - t
                passes = it + 1 if it < iters else iters
                runs.append(dt / (passes + 1))  # one residual product + `passes` loop products were timed
            return min(runs), passes, runs

        sec_it, passes, runs = timed(lambda m: O.cg_eigen(A, b, tol=1e-8, max_iter=m)[1], args.budget * 0.7)
        iter_bytes = 12 * nnz + 156 * A.n  # the unfused Eigen loop, which is what the faithful port runs
        out = {"value": A.n / (sec_it * (gpu_passes + 1)), "unit": "DOF/s", "cores": cores, "kind": "port",
               "variant": "reference-faithful (Eigen's unfused loop, OpenMP)",
               "sample": f"best of {REPEATS} x {passes} of {gpu_passes} PCG iterations, {N}^3, oracle.cg_eigen, {pin}, scaled",
               "host_cpu_quota": quota or None,
               "seconds_per_iteration": sec_it, "gbs": iter_bytes / sec_it / 1e9, "stream_triad_gbs": triad_gbs,
               "frac_of_stream_triad": (iter_bytes / sec_it / 1e9) / triad_gbs if triad_gbs > 0 else None}
        try:
            # (the iteration loop alone, as the function reports it: its private first-touch copies are a per-factorize cost)
            O.cg_jacobi_tuned(A, b, tol=1e-8, max_iter=2)
            _, it_p, _, sec_p = O.cg_jacobi_tuned(A, b, tol=1e-8, max_iter=8, loop_seconds=True)
            passes_t = int(max(8, min(gpu_passes, args.budget * 0.3 / REPEATS / max(sec_p / 8, 1e-6))))
            sec_t = 1e30
            for _ in range(REPEATS):
                _, it_t, _, sec = O.cg_jacobi_tuned(A, b, tol=1e-8, max_iter=passes_t, loop_seconds=True)
                sec_t = min(sec_t, sec / max(it_t + (1 if it_t < passes_t else 0), 1))
            tb = 12 * nnz + 100 * A.n
            out.update(tuned_value=A.n / (sec_t * (gpu_passes + 1)), tuned_seconds_per_iteration=sec_t,
                       tuned_gbs=tb / sec_t / 1e9, tuned_frac_of_stream_triad=(tb / sec_t / 1e9) / triad_gbs if triad_gbs > 0 else None,
                       tuned_variant=f"port-tuned: same recurrence, 3 fused passes (12 nnz + 100 n B), private first-touch copies; {passes_t} iterations timed")
        except Exception as e:
            out.update(tuned_value=None, tuned_variant=f"failed: {e}")
        # the reference's own build has no -fopenmp: Eigen::ConjugateGradient runs on ONE thread there
        O.lib().orc_set_num_threads(1)
        it1 = 4
        t = time.perf_counter()
        O.cg_eigen(A, b, tol=1e-8, max_iter=it1)
        dt1 = time.perf_counter() - t
        O.lib().orc_set_num_threads(cores)
        out["single_thread_value"] = A.n / (dt1 * (gpu_passes + 1) / (it1 + 1))
        print(json.dumps(out))
    else:
        REPEATS = int(os.environ.get("PSOLVE_BENCH_CPU_REPEATS", "3"))
        setups, solves = [], []
        for _ in range(REPEATS):  # best of three, setup and solve each
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
                                  "W-cycle, Chebyshev-16, 100 power iterations); OpenMP where AMGCL's builtin backend is, "
                                  "sequential where it is (aggregation sweep, transposes)"}))
    return 0


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


def _bytes_match(pmc, bytes_per_launch):
    """the vector kernels carry the same name with and without row kinds: a traffic file must also be for these bytes"""
    known = [pmc[k] for k in ("algorithmic_bytes", "stream_bytes", "csr_bytes") if pmc.get(k) is not None]
    return bytes_per_launch is None or not known or bytes_per_launch in known


def committed_pmc_traffic(kernel, world, N, precond, bytes_per_launch=None):
    """HBM traffic per launch of `kernel` from the COMMITTED rocprofv3 --pmc passes over this very command
    (scripts/evidence/pmc_bench.sh -> scripts/make_pmc_traffic.py), newest round first -- read from a file, not
    measured in this run; attached ONLY when the file was made for the instantiation the library reports."""
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
            pmc = json.load(open(f))
            if (world == 1 and N == 256 and precond == "jacobi" and pmc.get("workload") == "poisson7 256^3"
                    and kernel and pmc.get("kernel_library_name") == kernel
                    and _bytes_match(pmc, bytes_per_launch)):
                return pmc["traffic_bytes"], os.path.relpath(f, ROOT) + " (committed rocprofv3 --pmc passes of this command)"
    except Exception:
        pass
    return None, None


def live_pmc_traffic(kernel, args):
    """--live-traffic: HBM traffic per launch of `kernel` MEASURED IN THIS RUN -- two child processes, rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes) over one solve of this very
    command; FETCH_SIZE (KB) x 2 on gfx950, WRITE_SIZE in KB, means over the live launches.  None on any failure."""
    import csv, glob, shutil, subprocess, tempfile
    if not kernel or not shutil.which("rocprofv3"):
        return None, None
    tot = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="psolve_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "b", "--",
                   sys.executable, os.path.abspath(__file__), "--grid", str(args.grid), "--steps", "1", "--warmup", "0",
                   "--no-cpu-baseline", "--no-detail", "--precond", args.precond, "--storage", args.storage]
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=90, check=True)
            vals = []
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == counter and kernel in row["Kernel_Name"].replace("psolve::", ""):
                        vals.append(float(row["Counter_Value"]))
            shutil.rmtree(d, ignore_errors=True)
            live = [v for v in vals if v > 0.5 * max(vals)] if vals and max(vals) > 0 else []
            if not live:
                return None, None
            tot[counter] = sum(live) / len(live)
        return (tot["FETCH_SIZE"] * 1024.0 * 2.0 + tot["WRITE_SIZE"] * 1024.0,
                "measured in this run: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, two child passes over one solve (FETCH_SIZE x 2)")
    except Exception:
        return None, None


def storage_of(kernel: str) -> str:
    if kernel.startswith(("spmv_csr_kind", "spmv_csr_slots")):
        return "row_kinds"
    if kernel.startswith("spmv_csr_pat"):
        return "pattern_dictionary"
    return "plain_csr"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", type=int, default=256, help="N of the N^3 Poisson grid")
    ap.add_argument("--storage", default="csr", choices=["csr", "auto"],
                    help="csr (default): plain CSR stream, the contract kernel spmv_csr_dma; auto: what the backend picks for this "
                         "matrix (row kinds on this constant-coefficient grid: no matrix stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip both CPU legs (cpu_baseline and north_star's)")
    ap.add_argument("--no-detail", action="store_true",
                    help="only the timed region: no other storages, unstructured / elasticity / host-contract / north_star legs")
    ap.add_argument("--no-north-star", action="store_true", help="skip the 10 M-DOF AMG-PCG GPU-vs-CPU block of the detail")
    ap.add_argument("--live-traffic", action="store_true",
                    help="measure roofline.traffic in this run (two rocprofv3 --pmc child passes) instead of attaching the committed file")
    ap.add_argument("--elasticity-m", type=int, default=100, help="nodes per edge of the elasticity block (3 M^3 DOF)")
    ap.add_argument("--detail-file", default="bench_detail.json")
    ap.add_argument("--cpu-leg", default=None, choices=["eigen", "amgcl"], help=argparse.SUPPRESS)
    ap.add_argument("--passes", type=int, default=500, help=argparse.SUPPRESS)
    ap.add_argument("--budget", type=float, default=24.0, help=argparse.SUPPRESS)
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
    import bench_legs as legs

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
    csr = args.storage == "csr"
    s = HIPSolver("" if args.precond == "jacobi" else "Eigen::IdentityPreconditioner", device=local_rank)
    s.set_parameters({"HIP": {"tolerance": 1e-8, "max_iter": 20000, "profile_spmv": 8,
                              "spmv_kernel": 1 if csr else -1, "spmv_value_dict": not csr}})
    if args.precond == "amg":
        s.set_parameters({"HIP": {"precond": "amg", "amg": dict(legs.AMG_RECOMMENDED)}})
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
    if rank == 0 and not os.environ.get("PSOLVE_BENCH_NO_COPY"):
        # (torch's vectorised element-wise copy: 16 bytes of traffic per element.  Until round 6 this was the library's
        # scalar axpby on the vector kernels' small grid -- 3.1-3.7 TB/s, below what the SpMV itself moves, so not a
        # reference; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy)
        nc = 1 << 27
        try:
            tsrc = torch.zeros(nc, dtype=torch.float64, device=f"cuda:{local_rank}")
            tdst = torch.empty_like(tsrc)
            for _ in range(40):  # (the clocks of an idle device take milliseconds to come up)
                tdst.copy_(tsrc)
            torch.cuda.synchronize()
            copy_gbs = 0.0
            for _ in range(3):
                t1 = time.perf_counter()
                for _ in range(40):
                    tdst.copy_(tsrc)
                torch.cuda.synchronize()
                copy_gbs = max(copy_gbs, 40 * 16.0 * nc / (time.perf_counter() - t1) / 1e9)
            del tsrc, tdst
            torch.cuda.empty_cache()
        except Exception as e:  # (the reference point is optional: the line says null then)
            print(f"# device copy reference failed: {e}", file=sys.stderr)
            copy_gbs = None
    sync()
    box_before = legs.BoxSampler(local_rank)
    with box_before:  # (idle state right before the timed region: a few samples)
        time.sleep(0.1)
    sampler = legs.BoxSampler(local_rank)
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
        # the kernel of this line, as the LIBRARY reports it (psolve_hip_last_spmv_kernel: the instantiation PCG's product ran
        # on in the timed solves, spelled as rocprofv3 prints it)
        try:
            lib_kernel = s.last_spmv_kernel()
        except Exception:
            lib_kernel = ""
        npat, nkinds = int(s.get_param("spmv_patterns")), int(s.get_param("spmv_row_kinds"))
        stream_bytes, spmv_format = legs.spmv_stream_bytes(lib_kernel, n_loc, nnz_loc, npat, nkinds)
        storage = storage_of(lib_kernel)
        # The three kernels of a Jacobi-PCG iteration, each launched once per iteration and timed by HIP events inside the
        # timed solves (every 8th iteration, on the stream they are launched on); names from the library.  `roofline` is
        # the one that takes the most time per launch: on the CSR stream that is the product.
        kernels = [dict(legs.spmv_leg(lib_kernel or "unknown", stream_bytes, spmv_avg_ms, int(spmv_samples)), role="q = A p, p.q")]
        try:
            k2_ms, k3_ms = s.get_param("stats.update_r_ms_avg"), s.get_param("stats.update_xp_ms_avg")
            if k2_ms > 0 and k3_ms > 0:
                kd = int(s.get_param("pcg_kind_diag")) == 1  # (row kinds: 1 / diag read as table[kind[row]], 2 B per row)
                kernels.append(dict(legs.spmv_leg(s.last_pcg_kernel(1), (26 if kd else 32) * n_loc, k2_ms, int(spmv_samples)),
                                    role="r -= alpha q, r.r, r.z"))
                kernels.append(dict(legs.spmv_leg(s.last_pcg_kernel(2), (42 if kd else 48) * n_loc, k3_ms, int(spmv_samples)),
                                    role="x += alpha p, p = z + beta p"))
        except Exception:
            pass
        dom = max(kernels, key=lambda k: k["avg_launch_ms"])
        traffic, traffic_src = (live_pmc_traffic(dom["kernel"], args) if (args.live_traffic and world == 1) else (None, None))
        if traffic is None:
            traffic, traffic_src = committed_pmc_traffic(dom["kernel"], world, N, args.precond, dom["bytes_per_launch"])
        it_s = elapsed / args.steps / max(int(passes), 1)
        fused = sum(k["bytes_per_launch"] for k in kernels) if len(kernels) == 3 else stream_bytes + 80 * n_loc
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
            "config": {"workload": f"3-D 7-point Poisson {nx}x{ny}x{nz} ({n_global} DOF), {args.precond}-PCG to "
                                   f"||r||/||b||<1e-8, x0=0 (BASELINE.json configs[1])",
                       "storage": {"plain_csr": "plain CSR fp64/int32, matrix streamed every product (12 nnz + 20 n B)",
                                   "pattern_dictionary": "CSR values + 16-bit pattern id per row, no column stream",
                                   "row_kinds": "row kinds, no matrix stream (constant-coefficient grid only)"}[storage],
                       "precond": args.precond, "partition": f"{world} z-slab(s), {n_loc} rows + {n_halo} halo per GPU"},
            "iterations": int(passes),
            "ms_per_iteration": it_s * 1e3,
            "solver_error": info["solver_error"],
            "true_residual": info["true_residual"],
            # frac = bytes the kernel streams per launch / its in-loop launch time / 8 TB/s
            "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["achieved"] / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom["bytes_per_launch"],
                         "avg_launch_ms": dom["avg_launch_ms"], "launches_sampled": dom["launches_sampled"],
                         "device_copy_gbs_this_box": copy_gbs,
                         # (a reference only where it is one: on some boxes of this pool the 1 GiB -> 1 GiB copy runs at 3.1-3.7
                         # TB/s -- below every kernel of the solve -- while the solve's own rates are those of the other boxes)
                         "frac_of_device_copy": (dom["achieved"] / copy_gbs) if copy_gbs else None,  # (a read stream may beat a copy: > 1 is possible)
                         "iteration_bytes": fused, "iteration_gbs": fused / it_s / 1e9,
                         "iteration_frac": fused / it_s / 1e9 / HBM_PEAK_GBS},
            "comm_rccl_ranks_seen": int(s.get_param("dist.rccl_ranks_seen")),
        }
        detail = {"kernels": kernels, "spmv_format": spmv_format,
                  "eigen_unfused_bytes_per_iteration": 12 * nnz_loc + 156 * n_loc}
        if args.precond == "amg" and world == 1:  # the V-cycle's operations on levels 0 and 1, each against its bytes
            try:
                detail["amg_cycle_ops"] = legs.amg_cycle_ops(s, int(s.get_info()["amg_levels"]), block=False)
            except Exception as e:
                detail["amg_cycle_ops"] = {"failed": str(e)}
        if world > 1:  # per-iteration communication of rank 0, HIP events around the sampled iterations' collectives
            out["comm_allreduce_us_avg"] = s.get_param("stats.allreduce_us_avg")
            out["comm_halo_exchange_us_avg"] = s.get_param("stats.halo_us_avg")
        try:  # (after the timed region: latencies / gather rates of this box, probe.hip; 1 GiB of scratch)
            probe = s.box_probe()
        except Exception as e:
            probe = {"failed": str(e)}
        detail["box"] = dict(legs.box_static(local_rank), idle_before=box_before.summary(),
                             during_timed_region=sampler.summary(), probe=probe)
        sm = sampler.summary()
        out["box_sclk_mhz_median"] = (sm.get("sclk_mhz") or {}).get("median")
        out["box_power_w_median"] = (sm.get("power_w") or {}).get("median")

        # `value` by the storage the product ran on, side by side: the line's own and the same system on the two
        # storages a matrix with repeating rows gets (2 solves each, timed the same way)
        vbs = {storage: out["value"]}
        extra = world == 1 and args.precond == "jacobi" and not args.no_detail
        if extra:
            detail["storages"] = {}
            for name, prm in (("plain_csr", {"spmv_kernel": 1, "spmv_value_dict": False}),
                              ("pattern_dictionary", {"spmv_kernel": -1, "spmv_value_dict": False}),
                              ("row_kinds", {"spmv_kernel": -1, "spmv_value_dict": True})):
                if name in vbs:
                    continue
                try:
                    s.set_parameters({"HIP": prm})
                    s.generate_poisson7(nx, ny, nz, z0, z1)
                    dt, its, ms, smp, inf = legs.time_solves(s, b, x, n_loc, reps=2, warm_iters=32)
                    k = s.last_spmv_kernel()
                    if storage_of(k) != name:
                        continue  # (a grid too small for that storage: nothing to report under this name)
                    sb, _ = legs.spmv_stream_bytes(k, n_loc, nnz_loc, int(s.get_param("spmv_patterns")), int(s.get_param("spmv_row_kinds")))
                    tr, tr_src = committed_pmc_traffic(k, world, N, args.precond, sb)
                    detail["storages"][name] = legs.spmv_leg(k, sb, ms, smp, {
                        "iterations": its, "solve_s": dt, "dof_per_s": n_loc / dt, "ms_per_iteration": dt * 1e3 / max(its, 1),
                        "true_residual": inf["true_residual"], "traffic": tr, "traffic_source": tr_src})
                    vbs[name] = n_loc / dt
                except Exception as e:
                    detail["storages"][name] = {"failed": str(e)}
        out["value_by_storage"] = vbs
        if world == 1 and not args.no_cpu_baseline:
            try:
                cb = run_cpu_leg("eigen", grid=N, passes=int(passes))
                detail["cpu_baseline"] = cb
                out["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "variant", "sample", "gbs",
                                                              "stream_triad_gbs", "frac_of_stream_triad", "tuned_value", "tuned_gbs",
                                                              "tuned_frac_of_stream_triad", "single_thread_value")}
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "DOF/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"[:200]}
        else:
            out["cpu_baseline"] = None
        if world == 1:  # (a shard's communicator is torn down by every rank together, at exit)
            b.free()
            x.free()
            del s
        also = {}
        if extra:
            try:
                detail["unstructured"] = u = legs.unstructured_block(HIPSolver, N)
                also["poisson_random_numbering_dof_per_s"] = u["random"]["dof_per_s"]
                also["poisson_random_numbering_spmv_frac"] = u["random"]["frac"]
                also["poisson_windowed_numbering_dof_per_s"] = u["windowed_4096"]["dof_per_s"]
            except Exception as e:
                detail["unstructured"] = {"failed": str(e)}
            try:
                detail["elasticity"] = e = legs.elasticity_block(HIPSolver, args.elasticity_m)
                also.update(elasticity_dof=3 * args.elasticity_m ** 3, elasticity_solve_s=e["solve_s"], elasticity_iterations=e["iterations"],
                            elasticity_setup_s=e["generate_plus_setup_s"], elasticity_refresh_s=e["generate_plus_refresh_s"])
                r = e["unstructured"]["random_nodes"]
                also.update(elasticity_random_nodes_solve_s=r["solve_s"], elasticity_random_nodes_iterations=r["iterations"],
                            elasticity_random_nodes_setup_s=r["generate_plus_setup_s"],
                            elasticity_random_nodes_bsr3_spmv_frac=r["spmv"]["frac"],
                            elasticity_random_nodes_cheb_step_frac=r["cycle_ops"][0]["ops"]["cheb_step"]["frac_of_peak"])
                for name in ("matrix_fp32", "compact_direct", "compact_direct_matrix_fp32"):  # opt-in configurations, labelled as such
                    o = r.get("opt_in_" + name) or {}
                    if "solve_s" in o:
                        also[f"elasticity_random_nodes_opt_in_{name}_solve_s"] = o["solve_s"]
                        also[f"elasticity_random_nodes_opt_in_{name}_iterations"] = o["iterations"]
                        also[f"elasticity_random_nodes_opt_in_{name}_setup_s"] = o["generate_plus_setup_s"]
            except Exception as e:
                detail.setdefault("elasticity", {"failed": str(e)})
            try:  # what PolyFEM actually calls: host arrays in, host vectors out (PCIe inside the timed calls)
                detail["host_contract"] = h = legs.host_contract_block(HIPSolver, np, N, args.elasticity_m)
                for k in ("poisson", "elasticity"):
                    also[f"host_{k}_refactorize_s"] = h[k]["factorize_same_pattern"]["seconds"]
                    also[f"host_{k}_solve_s"] = h[k]["solve"]["seconds"]
            except Exception as e:
                detail.setdefault("host_contract", {"failed": str(e)})
            if N == 256 and not args.no_north_star:
                # the north_star's 10 M-DOF AMG-PCG comparison (GPU vs one CPU socket)
                try:
                    detail["north_star"] = ns = legs.north_star_block(HIPSolver, np, run_cpu_leg=None if args.no_cpu_baseline else run_cpu_leg)
                    g = ns["gpu_recommended_config"]
                    also.update(north_star_10m_dof_setup_s=g["setup_s"], north_star_10m_dof_solve_s=g["solve_s"],
                                north_star_reference_config_solve_s=ns["gpu_reference_config"]["solve_s"])
                    c = ns.get("cpu_amgcl_single_socket") or {}
                    if "solve_s" in c:
                        also.update(north_star_cpu_amgcl_port_setup_s=c["setup_s"], north_star_cpu_amgcl_port_solve_s=c["solve_s"],
                                    north_star_cpu_cores=c["cores"],
                                    north_star_speedup_setup_plus_solve=g.get("speedup_setup_plus_solve"),
                                    north_star_reference_config_speedup_setup_plus_solve=ns["gpu_reference_config"].get("speedup_setup_plus_solve"))
                except Exception as e:
                    detail.setdefault("north_star", {"failed": str(e)})
        if also:
            out["also"] = {k: (round(v, 6) if isinstance(v, float) else v) for k, v in also.items()}
        # the detail: a file next to this script (+ gpurun_out/ where it exists), a digest on stderr; the LINE goes last, alone
        full = dict(out, detail=detail)
        wrote = []
        for path in (os.path.join(ROOT, args.detail_file), os.path.join(ROOT, "gpurun_out", args.detail_file)):
            try:
                if os.path.isdir(os.path.dirname(path)):
                    with open(path, "w") as f:
                        json.dump(full, f, indent=1)
                    wrote.append(os.path.relpath(path, ROOT))
            except OSError:
                pass
        out["detail"] = wrote[0] if wrote else None
        for k, v in (out.get("also") or {}).items():
            print(f"# detail {k} = {v}", file=sys.stderr)
        sys.stderr.flush()
        line = json.dumps(out)
        assert len(line) < 8192, len(line)
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
