"""profiles/rNN_pmc_traffic_<name>.json from the PMC passes of scripts/evidence/pmc_bench.sh
(gpurun_out/rNN_bench_pmc_summary_<tag>.json): HBM bytes per launch of the kernels of a Jacobi-PCG iteration of the bench
system (Poisson 256^3), one file per kernel -- bench.py attaches the file whose `kernel_library_name` is the kernel its
roofline object names.  Corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE (KB) x 2 on gfx950 for wide coalesced
reads, cross-checked with TCC_EA0_RDREQ x 128 B; WRITE_SIZE in KB; means over the live launches of the solve.

    python scripts/evidence/make_pmc_traffic.py r06 csr     (tag: csr = --storage csr; auto = --storage auto, row kinds)"""
import json, re, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
tag = sys.argv[2] if len(sys.argv) > 2 else "csr"
S = json.load(open(f"gpurun_out/{rnd}_bench_pmc_summary_{tag}.json"))
n, nnz = 256 ** 3, 7 * 256 ** 3 - 6 * 256 ** 2


def pick(prefix):
    k = next(k for k in S if prefix in k)
    c = S[k]
    rd128 = c["TCC_EA0_RDREQ_sum"]["mean_live"] * 128.0
    rdfetch = c["FETCH_SIZE"]["mean_live"] * 1024.0 * 2.0
    wr = c["WRITE_SIZE"]["mean_live"] * 1024.0
    hit, miss = c["TCC_HIT_sum"]["mean_live"], c["TCC_MISS_sum"]["mean_live"]
    return k, dict(read_bytes_rdreq128=rd128, read_bytes_fetch_size_x2=rdfetch, write_bytes=wr, traffic_bytes=rd128 + wr,
                   l2_hit_rate=hit / (hit + miss), launches=c["FETCH_SIZE"]["n_live"])


if tag == "csr":   # the contract kernel and the two vector kernels beside it (1 / diag streamed: 32 n, 48 n)
    kernels = (("spmv_csr_dma<256, 1, double, true", 12 * nnz + 20 * n, "csr"), ("pcg_update_r_kernel", 32 * n, "csr_r"),
               ("pcg_update_xp_kernel", 48 * n, "csr_xp"))
else:              # row kinds: no matrix stream; 1 / diag by row kind (26 n, 42 n)
    kernels = (("spmv_csr_slots<1", 18 * n, "slots"), ("pcg_update_r_kernel", 26 * n, "r"), ("pcg_update_xp_kernel", 42 * n, "xp"))
cmd = f"python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-detail --storage {tag}"
for prefix, alg, short in kernels:
    k, v = pick(prefix)
    m = re.search(r"(spmv_\w+|pcg_\w+)<[^>]*>", k)
    out = {"workload": "poisson7 256^3", "kernel_library_name": m.group(0) if m else k, **v, "algorithmic_bytes": alg,
           "traffic_over_algorithmic": v["traffic_bytes"] / alg,
           "method": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum | TCC_HIT/MISS) over "
                     f"`{cmd}` (scripts/evidence/pmc_bench.sh {tag}); FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 reports half of "
                     "a wide coalesced read), cross-checked with TCC_EA0_RDREQ x 128 B; WRITE_SIZE in KB; means over the live "
                     "launches of the solve"}
    json.dump(out, open(f"profiles/{rnd}_pmc_traffic_{short}.json", "w"), indent=1)
    print(json.dumps(out, indent=1))
