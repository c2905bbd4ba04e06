"""one-line digest of a bench.py JSON line (tuning aid)"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
r = d["roofline"]
out = {"GN it/s": round(d["value"]), "us/iter": round(d["ms_per_step"] * 1e3, 2), "frac": round(r["frac"], 3),
       "iso_us": {k: round(v, 2) for k, v in d.get("kernels_isolated_avg_us", {}).items()}}
for key, field in (("tracker", "ms_per_frame"), ("depth_estimation", "gpu_ms_per_frame"), ("landmark_activation", "gpu_ms_per_keyframe"),
                   ("full_solve", "gpu_ms")):
    if d.get(key):
        out[key] = round(d[key][field], 4)
if d.get("roofline_large"):
    out["large_frac"] = round(d["roofline_large"]["frac"], 3)
    out["large_it/s"] = round(d["roofline_large"]["gn_iterations_per_s"])
if d.get("strong_scaling"):
    out["c3/c4 it/s"] = [round(d["strong_scaling"][k]["gn_iterations_per_s"]) for k in ("c3", "c4")]
print(json.dumps(out))
