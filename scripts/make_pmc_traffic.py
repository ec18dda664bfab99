"""profiles/rNN_pmc_traffic_<tag>.json from the PMC passes of scripts/gpu_r3_profiles.sh (gpurun_out/rNN_bench_pmc_summary_<tag>.json).
Corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE (KB) x 2 on gfx950 for wide coalesced reads, cross-checked
with TCC_EA0_RDREQ x 128 B; WRITE_SIZE in KB; means over the live launches of the solve.
    python scripts/make_pmc_traffic.py r03 pat | csr"""
import json, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
tag = sys.argv[2] if len(sys.argv) > 2 else "pat"
S = json.load(open(f"gpurun_out/{rnd}_bench_pmc_summary_{tag}.json"))
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
if tag == "kinds":
    # round 5, row kinds: the product streams no matrix; the iteration's longest kernel is pcg_update_xp.  One file per kernel
    # (bench.py attaches the file whose kernel_library_name is the kernel its roofline object names)
    import re
    cmd = "python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-extra"
    for prefix, alg, short in (("pcg_update_xp_kernel", 42 * n, "xp"), ("pcg_update_r_kernel", 26 * n, "r"), ("spmv_csr_slots<1", 18 * n, "slots")):
        k, v = pick(prefix)
        m = re.search(r"(spmv_\w+|pcg_\w+)<[^>]*>", k)
        out = {"workload": "poisson7 256^3", "kernel_library_name": m.group(0) if m else k, **v, "algorithmic_bytes": alg,
               "traffic_over_algorithmic": v["traffic_bytes"] / alg,
               "method": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum | TCC_HIT/MISS) over "
                         f"`{cmd}` (scripts/r5/evidence_kinds.sh); FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 reports half of a "
                         "wide coalesced read), cross-checked with TCC_EA0_RDREQ x 128 B; WRITE_SIZE in KB; means over the live "
                         "launches of the solve"}
        json.dump(out, open(f"profiles/{rnd}_pmc_traffic_{short}.json", "w"), indent=1)
        print(json.dumps(out, indent=1))
    sys.exit(0)
pat = tag == "pat"
k, sp = pick("spmv_csr_pat<256, 1, true>" if pat else "spmv_csr_dma<256, 1, double, true")
cmd = "python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-north-star --no-extra" + ("" if pat else " --spmv-kernel 1")
import re
_m = re.search(r"spmv_\w+<[^>]*>", k)
out = {"workload": "poisson7 256^3",
       # the instantiation as rocprofv3 prints it = what psolve_hip_last_spmv_kernel reports: bench.py attaches this file's
       # traffic only to a line whose kernel is this one (round 5)
       "kernel_library_name": _m.group(0) if _m else k,
       "kernel": "spmv_csr_pat<SPMV_DOT, nt>" if pat else "spmv_csr_dma<256, SPMV_DOT, double, nt>",
       "schedule": "xcd_map 2 (8192-row chunks dealt to the XCDs), LDS-DMA nt stream, nt y stores"
                   + ("; pattern dictionary, no column stream" if pat else ""), **sp,
       "csr_bytes": 12 * nnz + 20 * n, "stream_bytes": (8 * nnz + 22 * n) if pat else (12 * nnz + 20 * n),
       "method": "rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum | TCC_HIT/MISS) over "
                 f"`{cmd}` (scripts/r5/evidence.sh); FETCH_SIZE x2 "
                 "per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read), cross-checked with TCC_EA0_RDREQ x 128 B; "
                 "WRITE_SIZE in KB; means over the live launches of the solve",
       "other_kernels": {}}
for name, alg in (("pcg_update_r_kernel", 32 * n), ("pcg_update_xp_kernel", 48 * n)):
    kk, v = pick(name)
    out["other_kernels"][kk.replace("void psolve::", "")] = dict(traffic_bytes=v["traffic_bytes"], algorithmic_bytes=alg,
                                                                   l2_hit_rate=v["l2_hit_rate"])
json.dump(out, open(f"profiles/{rnd}_pmc_traffic_{tag}.json", "w"), indent=1)
print(json.dumps(out, indent=1))
