#!/bin/bash
# quick C1 check on the GPU box: core parity tests, per-kernel rocprof averages of the bench loop, three bench values
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_window_group.py -x -q -m gpu 2>&1 | tail -3
(cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_c1c && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c1c -o prof -- python $GRAFT_REPO_ROOT/scripts/profile_target.py c1 > /dev/null 2>&1)
f=$(find gpurun_out/prof_c1c -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Name"]).replace("dsopp_hip::", "").replace("void ", "")
    if "rocclr" in n or float(r["Percentage"]) < 0.5: continue
    print(f"{n[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:7.2f} us  {r['Percentage']}%")
PY
for i in 1 2 3; do python bench.py --no-cpu --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('GN it/s', round(d['value'],1), 'us/it', round(d['ms_per_step']*1e3,2))"; done
