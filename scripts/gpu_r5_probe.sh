#!/bin/bash
# round 5: atomics fan-in probe (geometry of the landmark-major linearisation) + parity of the solve-launch changes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 60 scripts/probes/bin/atomic_fanin_probe 2>&1 | tee gpurun_out/atomic_fanin_probe.txt
timeout 1200 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_window_group.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/pytest_partial.log
