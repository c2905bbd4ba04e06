"""Per-keyframe host-side breakdown of a rocprofv3 --hip-trace --kernel-trace --memory-copy-trace run of the native tick driver: HIP API
calls (count, total and mean duration), kernels and copies, divided by the number of keyframes / frames the driver reported."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, n_frames, n_keyframes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])


def table(pattern, name_col, start="Start_Timestamp", end="End_Timestamp"):
    files = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    agg = defaultdict(lambda: [0, 0.0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                a = agg[row[name_col]]
                a[0] += 1
                a[1] += (int(row[end]) - int(row[start])) / 1e3
    return agg


out = {"frames": n_frames, "keyframes": n_keyframes}
for key, pattern, col in (("hip_api", "*hip_api_trace.csv", "Function"), ("kernels", "*kernel_trace.csv", "Kernel_Name"), ("copies", "*memory_copy_trace.csv", "Direction")):
    agg = table(pattern, col)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]
    out[key] = [{"name": k[:110], "calls": v[0], "total_us": round(v[1], 1), "mean_us": round(v[1] / v[0], 2), "calls_per_frame": round(v[0] / n_frames, 2)} for k, v in rows]
    out[key + "_total_us_per_frame"] = round(sum(v[1] for v in agg.values()) / n_frames, 1)
json.dump(out, sys.stdout, indent=1)
