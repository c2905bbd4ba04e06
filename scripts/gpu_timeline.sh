#!/bin/bash
# kernel-by-kernel timeline of one solve of a workload: bash scripts/gpu_timeline.sh c1|c3_loop|large_loop  -> gpurun_out/<w>_one_solve_timeline.csv
cd $GRAFT_REPO_ROOT
w=${1:-c3_loop}
(cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_tl && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_tl -o prof -- python $GRAFT_REPO_ROOT/scripts/profile_target.py $w > /dev/null 2>&1)
f=$(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
python scripts/one_solve_timeline.py "$f" > gpurun_out/${w}_one_solve_timeline.csv
cat gpurun_out/${w}_one_solve_timeline.csv
