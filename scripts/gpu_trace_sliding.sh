#!/bin/bash
# kernel + copy timeline of the last keyframe step of scripts/time_sliding.py (push_frame .. marginalisation flags)
cd $GRAFT_REPO_ROOT
python scripts/time_sliding.py 2>/dev/null | tail -3
(cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_sl && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_sl -o prof -- python $GRAFT_REPO_ROOT/scripts/time_sliding.py > /dev/null 2>&1)
python - <<'PY'
import csv, glob, re
k = glob.glob("gpurun_out/prof_sl/**/*kernel_trace.csv", recursive=True)[0]
m = glob.glob("gpurun_out/prof_sl/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(k)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("dsopp_hip::", "")
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:70]))
if m:
    for r in csv.DictReader(open(m[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# last keyframe step: from the last foldIn-ish kernel; simply print the last 140 events
t0 = ev[-140][0]
prev = ev[-140][1]
for s, e, n in ev[-140:]:
    print(f"{(s - t0) / 1000:9.2f} us  +{(e - s) / 1000:7.2f}  gap {(s - prev) / 1000:7.2f}  {n}")
    prev = e
PY
