"""Host-side breakdown of a rocprofv3 --hip-trace --kernel-trace --memory-copy-trace run of the native tick driver: HIP API calls (count, total
and mean duration), kernels and copies per frame — and, with the driver's phase log (DSOPP_TICK_PHASE_LOG: CLOCK_MONOTONIC boundaries of every
frame / keyframe phase), the same per PHASE of the keyframe path: which calls fill activation_and_appends / push_frame / solve / ..."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, n_frames, n_keyframes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
phase_log = sys.argv[4] if len(sys.argv) > 4 else None


def rows(pattern, name_col, start="Start_Timestamp", end="End_Timestamp"):
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                yield row[name_col], int(row[start]), int(row[end])


out = {"frames": n_frames, "keyframes": n_keyframes}
api = list(rows("*hip_api_trace.csv", "Function"))
kernels = list(rows("*kernel_trace.csv", "Kernel_Name"))
copies = list(rows("*memory_copy_trace.csv", "Direction"))
for key, data in (("hip_api", api), ("kernels", kernels), ("copies", copies)):
    agg = defaultdict(lambda: [0, 0.0])
    for name, s, e in data:
        agg[name][0] += 1
        agg[name][1] += (e - s) / 1e3
    top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]
    out[key] = [{"name": k[:100], "calls": v[0], "total_us": round(v[1], 1), "mean_us": round(v[1] / v[0], 2), "calls_per_frame": round(v[0] / n_frames, 2)} for k, v in top]
    out[key + "_total_us_per_frame"] = round(sum(v[1] for v in agg.values()) / n_frames, 1)

if phase_log and os.path.exists(phase_log):
    # the tracer's timestamps and the driver's steady_clock are the same CLOCK_MONOTONIC when their ranges overlap: checked below
    kf_names = ["activation_and_appends", "push_frame", "solve", "update_frames", "marginalisation", "depth_maps"]
    fr_names = ["pyramid_object", "pyramid_build", "estimate_pose", "optical_flow", "depth_estimation"]
    intervals = []
    for ln in open(phase_log):
        v = ln.split()
        ts = [int(float(x) * 1e9) for x in v[2:]]
        names = kf_names if v[0] == "keyframe" else fr_names
        for i, nm in enumerate(names):
            intervals.append((ts[i], ts[i + 1], ("keyframe." if v[0] == "keyframe" else "frame.") + nm))
    intervals.sort()
    lo, hi = intervals[0][0], intervals[-1][1]
    inside = sum(1 for _, s, _e in api if lo <= s <= hi)
    out["phase_log"] = {"intervals": len(intervals), "api_calls_inside_the_logged_range": inside, "api_calls": len(api)}
    if inside > 0.5 * len(api):
        import bisect
        starts = [iv[0] for iv in intervals]
        per = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for name, s, e in api:
            j = bisect.bisect_right(starts, s) - 1
            if j >= 0 and s < intervals[j][1]:
                a = per[intervals[j][2]][name]
                a[0] += 1
                a[1] += (e - s) / 1e3
        kper = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for name, s, e in kernels:   # kernels by the phase they RAN in
            j = bisect.bisect_right(starts, s) - 1
            if j >= 0 and s < intervals[j][1]:
                a = kper[intervals[j][2]][name.split("(")[0][-60:]]
                a[0] += 1
                a[1] += (e - s) / 1e3
        phases = {}
        for ph, agg in per.items():
            div = n_keyframes if ph.startswith("keyframe.") else n_frames
            top = sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]
            phases[ph] = {"host_us_in_hip_calls_per_occurrence": round(sum(v[1] for v in agg.values()) / div, 1),
                          "hip_calls_per_occurrence": round(sum(v[0] for v in agg.values()) / div, 1),
                          "top_calls": [{"name": k, "per_occurrence": round(v[0] / div, 1), "us_per_occurrence": round(v[1] / div, 1)} for k, v in top],
                          "kernel_us_per_occurrence": round(sum(v[1] for v in kper[ph].values()) / div, 1),
                          "kernels_per_occurrence": round(sum(v[0] for v in kper[ph].values()) / div, 1),
                          "top_kernels": [{"name": k, "per_occurrence": round(v[0] / div, 1), "us_per_occurrence": round(v[1] / div, 1)}
                                          for k, v in sorted(kper[ph].items(), key=lambda kv: -kv[1][1])[:6]]}
        out["by_phase"] = phases
json.dump(out, sys.stdout, indent=1)
