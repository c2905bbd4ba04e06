#!/bin/bash
# one GPU-box session: parity tests, bench line, rocprof kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
timeout 600 python bench.py 2>gpurun_out/bench.err | tee gpurun_out/bench.json
tail -5 gpurun_out/bench.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 70 --no-cpu --no-extras > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2>$GRAFT_REPO_ROOT/gpurun_out/rocprof.err
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/rocprof.err
find gpurun_out/prof -name "*stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f"
