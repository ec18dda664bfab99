"""The driver's bench contract, on a small grid: one JSON line with the agreed keys (bench.py docstring), values
that make sense, for the default Jacobi configuration and for --precond amg."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra, tmp_path):
    detail = str(tmp_path / "detail.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--grid", "48", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--elasticity-m", "12", "--detail-file", detail] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.splitlines()
    assert len([l for l in lines if l.startswith("{")]) == 1 and lines[-1].startswith("{")  # one JSON line, and it is the LAST one
    line = lines[-1]
    # the driver's record must parse (round 5's 23 KB line did not): a few KB, flat, scalars only inside roofline / cpu_baseline
    assert len(line) < 8192
    j = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    scalar = (str, int, float, bool, type(None))
    assert all(isinstance(v, scalar) for v in j["roofline"].values())
    assert all(isinstance(v, scalar) for v in j["config"].values()) and set(j["config"]) == {"workload", "storage", "precond", "partition"}
    assert all(isinstance(v, scalar) for v in (j.get("also") or {}).values())
    assert j["unit"] == "DOF/s" and j["dtype"] == "f64" and j["data"] == "synthetic" and j["higher_is_better"] is True
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["vs_baseline"] is None
    assert j["value"] > 0 and abs(j["value"] - 48 ** 3 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["frac"] <= 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert 0 < r["iteration_frac"] <= 1.0 and r["launches_sampled"] > 0
    assert j["true_residual"] < 1.5e-8 and j["iterations"] > 0 and j["comm_rccl_ranks_seen"] == 0, \
        (j["true_residual"], j["iterations"], j.get("solver_error"), j["ms_per_step"], out.stderr[-3000:])
    d = json.load(open(detail))
    assert d["value"] == j["value"] and j["detail"].endswith("detail.json")
    return j, d["detail"]


def test_bench_line_is_the_contract_csr_kernel(tmp_path):
    """Default storage: the matrix streamed as plain CSR -- the roofline object is the north_star's kernel on SURVEY.md 8(d)'s
    12 nnz + 20 n bytes; the storages a constant-coefficient grid collapses to are named special cases beside it."""
    j, d = _run_bench([], tmp_path)
    n, nnz = 48 ** 3, 7 * 48 ** 3 - 6 * 48 ** 2
    r = j["roofline"]
    assert j["config"]["storage"].startswith("plain CSR")
    ks = d["kernels"]  # the kernels of the iteration, named by the LIBRARY; `roofline` is the one with the longest sampled launch
    assert ks[0]["kernel"].startswith("spmv_csr_") and "<" in ks[0]["kernel"] and ks[0]["bytes_per_launch"] == 12 * nnz + 20 * n
    assert not ks[0]["kernel"].startswith(("spmv_csr_slots", "spmv_csr_kind", "spmv_csr_pat"))
    assert r["kernel"] in [k["kernel"] for k in ks] and r["avg_launch_ms"] == max(k["avg_launch_ms"] for k in ks)
    assert len(ks) == 3 and ks[1]["kernel"].startswith("pcg_update_r_kernel<") and ks[2]["kernel"].startswith("pcg_update_xp_kernel<")
    assert ks[1]["bytes_per_launch"] == 32 * n and ks[2]["bytes_per_launch"] == 48 * n
    assert all(0 < k["frac"] <= 1.0 and k["avg_launch_ms"] > 0 for k in ks)
    vbs = j["value_by_storage"]
    assert set(vbs) == {"row_kinds", "pattern_dictionary", "plain_csr"} and vbs["plain_csr"] == j["value"]
    st = d["storages"]
    assert st["row_kinds"]["kernel"].startswith("spmv_csr_slots<1,") and st["row_kinds"]["bytes_per_launch"] == 18 * n
    assert st["pattern_dictionary"]["kernel"].startswith("spmv_csr_pat<256, 1,")
    for v in st.values():
        assert abs(v["iterations"] - j["iterations"]) <= 1 and 0 < v["frac"] <= 1.0 and v["true_residual"] < 1.5e-8
    # the detail legs: the unstructured renumberings (no dictionary), BASELINE.json configs[2], the host contract, the box
    for name in ("windowed_4096", "random"):
        u = d["unstructured"][name]
        assert u["patterns"] == 0 and 0 < u["frac"] <= 1.0 and u["true_residual"] < 1.5e-8
        assert abs(u["iterations"] - j["iterations"]) <= 3  # the same operator, renumbered
        c = u["caller_numbering"]  # "reorder" 0 next to the default
        assert c["reordered"] is False and 0 < c["frac"] <= 1.0 and abs(c["iterations"] - u["iterations"]) <= 2
        assert u["reordered"] == (48 ** 3 >= 131072)  # auto: small systems keep the caller's numbering
    e = d["elasticity"]
    assert e["iterations"] > 0 and e["true_residual"] < 1.5e-8 and 0 < e["spmv"]["frac"] <= 1.0
    assert e["direct_coarse"]["iterations"] <= e["iterations"]
    eu = e["unstructured"]["random_nodes"]  # the same matrix, nodes renumbered: the same blocks, about the same counts
    # (M = 12: 28 block-row kinds of 1728 nodes -- the product streams no matrix: kinds, x, y)
    assert e["spmv"]["kernel"].startswith("spmv_bsr3_kind<1,") and e["spmv"]["block_row_kinds"] == 28
    assert e["spmv"]["bytes_per_launch"] == 50 * e["spmv"]["block_rows"]
    assert eu["spmv"]["bytes_per_launch"] == 76 * eu["spmv"]["blocks"] + 52 * eu["spmv"]["block_rows"] and eu["spmv"]["block_row_kinds"] == 0
    assert e["reordered"] is False  # the generator's grid numbering stays
    assert eu["spmv"]["blocks"] == e["spmv"]["blocks"] and eu["true_residual"] < 1.5e-8
    assert eu["caller_numbering"]["reordered"] is False and eu["caller_numbering"]["true_residual"] < 1.5e-8
    ops = e["cycle_ops"][0]["ops"]
    assert set(ops) >= {"cheb_step", "residual", "restrict", "prolong", "cheb_first"}
    assert all(v["us"] > 0 and 0 < v["frac_of_peak"] <= 1.0 for v in ops.values())
    hc = d["host_contract"]
    for k in ("poisson", "elasticity"):
        assert hc[k]["pattern_uploads"] == 1 and hc[k]["factorize_same_pattern"]["h2d_gb"] < 0.7 * hc[k]["factorize_first"]["h2d_gb"]
        assert hc[k]["solve"]["true_residual"] < 1.5e-8
    pr = d["box"]["probe"]
    assert 10 < pr["latency_ns_l2_1mib"] < pr["latency_ns_hbm_1gib"] < 5000 and pr["ggathers_per_s_2mib"] > pr["ggathers_per_s_64mib"] > 0
    assert 500 < pr["shader_counter_mhz_under_load"] < 4000
    a = j["also"]  # the digest of the detail on the line: scalars
    assert a["elasticity_solve_s"] == round(e["solve_s"], 6) and a["elasticity_random_nodes_iterations"] == eu["iterations"]


def test_bench_line_storage_auto_and_amg(tmp_path):
    """--storage auto: the backend's own pick for this constant-coefficient grid (row kinds: no matrix stream, the longest
    launch is a vector update); --precond amg: the cycle's operations per level in the detail."""
    j, d = _run_bench(["--storage", "auto", "--no-detail"], tmp_path)
    n = 48 ** 3
    ks = d["kernels"]
    assert j["config"]["storage"].startswith("row kinds") and set(j["value_by_storage"]) == {"row_kinds"}
    assert ks[0]["kernel"].startswith("spmv_csr_slots<1,") and ks[0]["bytes_per_launch"] == 18 * n
    assert ks[1]["bytes_per_launch"] == 26 * n and ks[2]["bytes_per_launch"] == 42 * n  # (1 / diag by row kind)
    j, d = _run_bench(["--precond", "amg"], tmp_path)
    ops = d["amg_cycle_ops"][0]["ops"]
    assert all(v["us"] > 0 and 0 < v["frac_of_peak"] <= 1.0 for v in ops.values())


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` is its own launcher; on a node with fewer than N GPUs it must fail loudly
    instead of printing a line for fewer devices (round 1 silently ran one rank)."""
    import ctypes as C
    from polysolve_amd import _lib
    c = C.c_int()
    _lib.load().psolve_hip_device_count(C.byref(c))
    n = c.value + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--grid", "32", "--steps",
                          "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                         cwd=ROOT, env=env)
    assert out.returncode != 0
    assert "GPU(s) visible" in out.stderr and not any(l.startswith("{") for l in out.stdout.splitlines())
    # a launcher that started the wrong number of ranks is refused too
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "32", "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr
