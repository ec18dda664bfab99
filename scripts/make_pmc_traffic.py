"""profiles/rNN_pmc_traffic.json from the PMC passes of scripts/gpu_pmc_bench.sh (gpurun_out/bench_pmc_summary.json).
Corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE (KB) x 2 on gfx950 for wide coalesced reads, cross-checked
with TCC_EA0_RDREQ x 128 B; WRITE_SIZE in KB; means over the live launches of the solve."""
import json, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
S = json.load(open("gpurun_out/bench_pmc_summary.json"))
def pick(prefix):
    k = next(k for k in S if prefix in k)
    c = S[k]
    rd128 = c["TCC_EA0_RDREQ_sum"]["mean_live"] * 128.0
    rdfetch = c["FETCH_SIZE"]["mean_live"] * 1024.0 * 2.0
    wr = c["WRITE_SIZE"]["mean_live"] * 1024.0
    hit, miss = c["TCC_HIT_sum"]["mean_live"], c["TCC_MISS_sum"]["mean_live"]
    return k, dict(read_bytes_rdreq128=rd128, read_bytes_fetch_size_x2=rdfetch, write_bytes=wr, traffic_bytes=rd128 + wr,
                   l2_hit_rate=hit / (hit + miss), launches=c["FETCH_SIZE"]["n_live"])
n, nnz = 256 ** 3, 7 * 256 ** 3 - 6 * 256 ** 2
pat = any("spmv_csr_pat<1, true>" in k for k in S)
k, sp = pick("spmv_csr_pat<1, true>" if pat else "spmv_csr_dma<256, 1, double, true>")
out = {"workload": "poisson7 256^3",
       "kernel": "spmv_csr_pat<SPMV_DOT, nt>" if pat else "spmv_csr_dma<256, SPMV_DOT, double, nt>",
       "schedule": "xcd_map 2 (8192-row chunks dealt to the XCDs), LDS-DMA nt stream, nt y stores"
                   + ("; pattern dictionary, no column stream" if pat else ""), **sp,
       "algorithmic_bytes": 12 * nnz + 20 * n, "stream_bytes": (8 * nnz + 22 * n) if pat else (12 * nnz + 20 * n),
       "method": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum | TCC_HIT/MISS) over "
                 "`python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-north-star` (scripts/gpu_pmc_bench.sh); FETCH_SIZE x2 "
                 "per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read), cross-checked with TCC_EA0_RDREQ x 128 B; "
                 "WRITE_SIZE in KB; means over the live launches of the solve",
       "other_kernels": {}}
for name, alg in (("pcg_update_r_kernel", 32 * n), ("pcg_update_xp_kernel", 48 * n)):
    kk, v = pick(name)
    out["other_kernels"][kk.replace("void psolve::", "")] = dict(traffic_bytes=v["traffic_bytes"], algorithmic_bytes=alg,
                                                                   l2_hit_rate=v["l2_hit_rate"])
json.dump(out, open(f"profiles/{rnd}_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
