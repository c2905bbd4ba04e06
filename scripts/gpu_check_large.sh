#!/bin/bash
# after a change of the large-window / two-stage path: parity tests that cover it, then per-iteration times of C1 / C3 / C4
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_window_group.py tests/test_gpu_c4_end_to_end.py tests/test_gpu_distributed.py tests/test_gpu_masks.py tests/test_gpu_sliding_window.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python scripts/threshold_sweep.py 7 2000 2>/dev/null | grep "us per"
python scripts/threshold_sweep.py 7 20000 2>/dev/null | grep "us per"
python scripts/threshold_sweep.py 12 50000 1 2>/dev/null | grep "us per"
python scripts/shard_cost.py c3 2>/dev/null | grep "^{" > gpurun_out/shard_cost_c3.json; cat gpurun_out/shard_cost_c3.json
python scripts/shard_cost.py c4 2>/dev/null | grep "^{" > gpurun_out/shard_cost_c4.json; cat gpurun_out/shard_cost_c4.json
