#!/bin/bash
# round 6: the whole GPU suite + the driver's bench invocation + the tracker A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r06/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r06/bench_stdout.json 2> gpurun_out/r06/bench_stderr.log; echo "bench rc $?"; wc -c gpurun_out/r06/bench_stdout.json; cp bench_extras.json gpurun_out/r06/bench_extras.json
head -c 4200 gpurun_out/r06/bench_stdout.json; echo
LIBS="lib lib_exp_nosched" bash scripts/gpu_r6_tracker.sh 2>&1 | grep -v "^{" | tail -12
