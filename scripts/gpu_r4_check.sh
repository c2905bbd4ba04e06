#!/bin/bash
# round 4: parity tests of the window path + per-iteration times of C1 / C3 / C4 (scripts/time_large.py)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_masks.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -8
for cfg in "7 2000" "7 20000" "12 50000"; do timeout 300 python scripts/time_large.py $cfg 2>&1 | tail -1; done
