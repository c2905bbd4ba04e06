#!/bin/bash
# round 5: what caps concurrent windows — hardware queues (GPU_MAX_HW_QUEUES 4 / 8), enqueueing threads (one / one per window), traces
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/concurrent
mkdir -p $O
export TMPDIR=/tmp
{
for q in 2 4 8; do
  for n in 1 2 4 8 16; do
    for mode in threads async; do
      echo "GPU_MAX_HW_QUEUES=$q: $(GPU_MAX_HW_QUEUES=$q timeout 120 python scripts/concurrent_trace.py run $n $mode 700 2>&1 | tail -1)"
    done
  done
done
} | tee $O/rates.txt
for q in 4 8; do
  d=/tmp/prof_conc_$q
  rm -rf $d
  (cd /tmp && GPU_MAX_HW_QUEUES=$q timeout 200 rocprofv3 --kernel-trace --output-format csv -d $d -o prof -- python $GRAFT_REPO_ROOT/scripts/concurrent_trace.py run 8 threads 350 > $GRAFT_REPO_ROOT/$O/trace_run_q$q.log 2>&1)
  t=$(find $d -name '*kernel_trace.csv' | head -1)
  echo "== GPU_MAX_HW_QUEUES=$q, 8 windows, one thread each (under the profiler: $(tail -1 $O/trace_run_q$q.log))"
  [ -n "$t" ] && python scripts/concurrent_trace.py analyse "$t"
done | tee $O/trace_analysis.txt
