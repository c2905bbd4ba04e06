#!/bin/bash
# round 5: solve launch without scratch (packed marginal prior, batch of 6 at 512 threads, tail-only operands requested behind the
# factorisation) + native tick-sequence driver: parity, then per-iteration times at the window sizes the 512-thread kernel serves
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_masks.py tests/test_golden.py tests/test_gpu_degenerate.py tests/test_gpu_sliding_window.py tests/test_marginalization.py tests/test_gpu_window_group.py tests/test_gpu_tick_sequence.py tests/test_host_adapter.py -x -q -m gpu 2>&1 | tail -12 | tee $O/pytest_solver.log
for cfg in "7 2000" "7 20000" "12 50000" "15 5000" "12 8000" "16 6000"; do timeout 300 python scripts/time_large.py $cfg 2>&1 | tail -1; done | tee $O/time_windows.txt
