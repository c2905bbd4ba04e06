#!/bin/bash
# round 5: internal spatial order of the landmarks behind the C-ABI — parity of everything, then A/B on the large windows, then the
# one-solve kernel timeline of the large loop (random caller order, the product's internal order)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee $O/pytest_order.log
for rep in 1 2; do
for env in "DSOPP_HIP_LANDMARK_ORDER=caller" "DSOPP_HIP_LANDMARK_ORDER=sorted" "DSOPP_HIP_LANDMARK_ORDER=sorted DSOPP_HIP_SWEEP_XCD_BANDS=1"; do
  for cfg in "12 50000" "7 20000" "15 5000"; do echo "$env: $(env $env timeout 300 python scripts/time_large.py $cfg 2>&1 | tail -1)"; done
done; done | tee $O/time_landmark_order.txt
for what in large_loop; do
  d=/tmp/prof_$what
  rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o prof -- python $GRAFT_REPO_ROOT/scripts/profile_target.py $what > $GRAFT_REPO_ROOT/$O/$what.log 2>&1)
  t=$(find $d -name '*kernel_trace.csv' | head -1)
  s=$(find $d -name '*kernel_stats.csv' | head -1)
  [ -n "$t" ] && python scripts/one_solve_timeline.py "$t" > $O/${what}_one_solve_timeline.csv
  [ -n "$s" ] && cp "$s" $O/${what}_kernel_stats.csv
  tail -4 $O/${what}_one_solve_timeline.csv
done
