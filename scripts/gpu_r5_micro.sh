#!/bin/bash
# round 5: reduce launch prologue (control block words + descriptor head in one scalar round trip): parity subset, bench line, C1 per-kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_invariances.py tests/test_golden.py tests/test_gpu_sliding_window.py tests/test_marginalization.py tests/test_gpu_window_group.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_micro.log
for i in 1 2 3; do timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_micro_$i.json; python scripts/bench_brief.py $O/bench_micro_$i.json; done | tee $O/bench_micro.txt
