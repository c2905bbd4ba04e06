#!/bin/bash
# A/B: calculateIdepths inside the solve launch (default) against the round-3 flow (DSOPP_HIP_K3_BACKSUB=0: a kernel of its own in front
# of the sweeps of large windows, fused into the sweeps of small ones)
cd $GRAFT_REPO_ROOT
if [ "$1" != notests ]; then
timeout 1200 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_c4_end_to_end.py tests/test_gpu_degenerate.py tests/test_gpu_window_group.py tests/test_gpu_distributed.py tests/test_gpu_sliding_window.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|error" | tail -5
fi
for cfg in "7 2000" "7 20000" "12 50000" "9 30000" "16 30000" "15 5000" "12 100000"; do
  for v in 1 0; do
    echo -n "k3_backsub=$v  "; DSOPP_HIP_K3_BACKSUB=$v timeout 300 python scripts/time_large.py $cfg 2>&1 | grep -v amdgpu | tail -1
  done
done
