"""the headline numbers of bench.py in one line (no CPU legs, no extra blocks)"""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-north-star", "--no-extra", "--steps", "5", "--warmup", "1"] + args,
                     capture_output=True, text=True)
try:
    j = json.loads(out.stdout.strip().splitlines()[-1])
    r = j["roofline"]
    print(json.dumps(dict(mdofs=round(j["value"] / 1e6, 2), ms_step=round(j["ms_per_step"], 2), its=j["iterations"], ms_it=round(j["ms_per_iteration"], 4),
                          spmv_ms=round(r["avg_launch_ms"], 4), frac=round(r["frac"], 4), kernel=r["kernel"], copy=round(r["device_copy_gbs_this_box"] or 0),
                          it_frac=round(j["iteration_roofline"]["fused_frac_of_peak"], 4), sclk=j["box"]["during_timed_region"]["sclk_mhz"], pw=j["box"]["during_timed_region"]["power_w"])))
except Exception as e:
    print("failed", e, out.stdout[-500:], out.stderr[-1500:])
