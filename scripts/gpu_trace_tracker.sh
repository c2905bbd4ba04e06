#!/bin/bash
# kernel + copy timeline of the last two tracked frames of the tracker timing (profile_target.py tracker)
cd $GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_trk && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_trk -o prof -- python $GRAFT_REPO_ROOT/scripts/profile_target.py tracker > /dev/null 2>&1)
python - <<'PY'
import csv, glob, re
k = glob.glob("gpurun_out/prof_trk/**/*kernel_trace.csv", recursive=True)[0]
m = glob.glob("gpurun_out/prof_trk/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(k)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("dsopp_hip::", "")
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:60]))
if m:
    for r in csv.DictReader(open(m[0])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
idx = [i for i, e in enumerate(ev) if e[2].startswith("alignPyramidKernel")]
a = idx[-3] - 6
t0 = ev[a][0]; prev = ev[a][1]
for s, e, n in ev[a:idx[-1] + 4]:
    print(f"{(s - t0) / 1000:9.2f} us  +{(e - s) / 1000:7.2f}  gap {(s - prev) / 1000:7.2f}  {n}")
    prev = e
PY
