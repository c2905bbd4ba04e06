#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 70 --no-cpu --no-extras > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2>$GRAFT_REPO_ROOT/gpurun_out/rocprof.err
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cat "$f" | cut -c1-200 | head -30
