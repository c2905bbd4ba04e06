#!/bin/bash
# kernel-by-kernel timeline of one 12 KF / 50 000-point solve + the per-iteration time of that window
cd $GRAFT_REPO_ROOT
(cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_c4 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_c4 -o prof -- python $GRAFT_REPO_ROOT/scripts/profile_target.py large_loop > /dev/null 2>&1)
f=$(find gpurun_out/prof_c4 -name "*kernel_trace.csv" | head -1)
python scripts/one_solve_timeline.py "$f" > gpurun_out/c4_one_solve_timeline.csv
cat gpurun_out/c4_one_solve_timeline.csv
python scripts/shard_cost.py c4 2>&1 | grep -v amdgpu | tail -5
timeout 900 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_window_group.py tests/test_gpu_c4_end_to_end.py tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
