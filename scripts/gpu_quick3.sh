#!/bin/bash
# quick check after a solve / loop change: core parity tests + per-iteration times of C1 / C3 / C4 (same harness as threshold_sweep.py)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_window_group.py tests/test_gpu_distributed.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for i in 1 2; do
python scripts/threshold_sweep.py 7 2000 2>/dev/null | grep "us per"
python scripts/threshold_sweep.py 7 20000 2>/dev/null | grep "us per"
python scripts/threshold_sweep.py 12 50000 1 2>/dev/null | grep "us per"
done
