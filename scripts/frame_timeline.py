"""One tracked frame of the native tick driver as a timeline: the driver's phase boundaries (DSOPP_TICK_PHASE_LOG) and every kernel / copy / blocking HIP call
of a rocprofv3 --hip-trace --kernel-trace --memory-copy-trace run between them, in microseconds from the frame's start.
   python scripts/frame_timeline.py <rocprof output dir> <phase log> [frame | keyframe] [index, default: one in the middle]"""
import csv
import glob
import os
import sys

d, phase_log = sys.argv[1], sys.argv[2]


def rows(pattern, name_col):
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                yield row[name_col], int(row["Start_Timestamp"]), int(row["End_Timestamp"])


mode = sys.argv[3] if len(sys.argv) > 3 else "frame"
frames = [ln.split() for ln in open(phase_log) if ln.startswith("frame")]
kfs = [ln.split() for ln in open(phase_log) if ln.startswith("keyframe")]
keyframes = {k[1] for k in kfs}
plain = [f for f in frames if f[1] not in keyframes] if mode == "frame" else kfs
pick = plain[int(sys.argv[4])] if len(sys.argv) > 4 else plain[len(plain) // 2]
ts = [int(float(x) * 1e9) for x in pick[2:]]
names = ["pyramid_object", "pyramid_build", "estimate_pose", "optical_flow", "depth_estimation"] if mode == "frame" else \
    ["activation_and_appends", "push_frame", "solve", "update_frames", "marginalisation", "depth_maps"]
t0, t1 = ts[0], ts[-1]
events = []
for i, nm in enumerate(names):
    events.append((ts[i], ts[i + 1], "PHASE", nm))
for kind, pattern, col in (("kernel", "*kernel_trace.csv", "Kernel_Name"), ("copy", "*memory_copy_trace.csv", "Direction"), ("api", "*hip_api_trace.csv", "Function")):
    for name, s, e in rows(pattern, col):
        if e < t0 or s > t1:
            continue
        if kind == "api" and (e - s) < 2000 and not name.startswith(("hipMemcpy", "hipLaunch")):
            continue
        short = name.split("(")[0].replace("dsopp_hip::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:48]
        events.append((s, e, kind, short))
events.sort()
print(f"{mode} {pick[1]}: {(t1 - t0) / 1e3:.1f} us")
for s, e, kind, name in events:
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f} us)  {kind:6s} {name}")
