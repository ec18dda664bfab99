"""What the GPU box's host gives the CPU baseline: cgroup quota, topology, and the two CG legs by thread count."""
import os, sys, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    nt = int(sys.argv[1])
    import oracle as O
    A = O.poisson7(256)
    b = O.spmv(A, O.splitmix_vector(A.n, 42))
    tri = O.stream_triad(1 << 25, 5)
    out = {"threads": nt, "triad_gbs": tri}
    for name, fn, B in (("eigen", lambda m: O.cg_eigen(A, b, max_iter=m), 12 * A.nnz + 156 * A.n),
                        ("tuned", lambda m: O.cg_jacobi_tuned(A, b, max_iter=m), 12 * A.nnz + 100 * A.n)):
        fn(2)
        best = 1e30
        for _ in range(2):
            t = time.perf_counter(); fn(20); best = min(best, (time.perf_counter() - t) / 21)
        out[name + "_ms_it"] = best * 1e3
        out[name + "_gbs"] = B / best / 1e9
    t = time.perf_counter()
    for _ in range(10): O.spmv(A, b)
    out["spmv_ms"] = (time.perf_counter() - t) * 100
    t = time.perf_counter()
    for _ in range(10): O.lib().orc_dot(A.n, b, b)
    out["dot_ms"] = (time.perf_counter() - t) * 100
    print(json.dumps(out))
    sys.exit(0)
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpu.stat"):
    try: print(f, open(f).read().strip().replace("\n", " | "))
    except OSError as e: print(f, "-", e)
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA|MHz' ; numactl -H 2>/dev/null | head -20; cat /proc/loadavg")
for nt, bind in ((16, "close"), (16, "spread"), (24, "spread"), (32, "spread")):
    env = dict(os.environ, OMP_NUM_THREADS=str(nt), OMP_PLACES="{0}:64", OMP_PROC_BIND=bind)
    print(nt, bind)
    r = subprocess.run([sys.executable, __file__, str(nt)], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-300:])
    try: print("  cpu.stat", open("/sys/fs/cgroup/cpu.stat").read().strip().replace("\n", " | "))
    except OSError: pass
