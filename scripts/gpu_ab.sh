#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
python scripts/threshold_sweep.py 7 2000 2>/dev/null | grep "us per"
DSOPP_HIP_DECIDE_IN_SOLVE=1 python scripts/threshold_sweep.py 7 2000 2>/dev/null | grep "us per" | sed 's/$/  (decide in solve)/'
done
python scripts/threshold_sweep.py 7 8000 2>/dev/null | grep "us per"
DSOPP_HIP_DECIDE_IN_SOLVE=1 python scripts/threshold_sweep.py 7 8000 2>/dev/null | grep "us per" | sed 's/$/  (decide in solve)/'
DSOPP_HIP_DECIDE_IN_SOLVE=1 timeout 600 python -m pytest tests/test_gpu_pba.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
