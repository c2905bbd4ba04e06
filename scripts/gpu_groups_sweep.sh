#!/bin/bash
# tuning aid: per-iteration time against the number of item groups per sweep workgroup (DSOPP_HIP_SWEEP_GROUPS)
cd $GRAFT_REPO_ROOT
for P in 8000 20000 30000; do for g in 0 2 3 4 5 6 8; do echo -n "groups $g: "; DSOPP_HIP_SWEEP_GROUPS=$g python scripts/threshold_sweep.py 7 $P 2>/dev/null | grep "us per"; done; done
for g in 0 4 5 6 8 10; do echo -n "groups $g: "; DSOPP_HIP_SWEEP_GROUPS=$g python scripts/threshold_sweep.py 12 50000 1 2>/dev/null | grep "us per"; done
