#!/bin/bash
# depth-estimation parity tests + kernel time under rocprofv3
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_depth_estimation.py tests/test_gpu_tick_sequence.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pd && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o prof -- python $GRAFT_REPO_ROOT/scripts/profile_target.py depth > /tmp/pd.log 2>&1)
grep -h "estimateDepths" $(find /tmp/pd -name "*kernel_stats.csv") | sed 's/.*DepthLandmarks const\*)",//' 
tail -1 /tmp/pd.log | cut -c1-140
