#!/bin/bash
# experiment: copies of the combined system the reduction launch spreads its atomics over (DSOPP_HIP_COMB_COPIES = 1 / 2 / 4)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
out=$O/time_comb_copies_ab.txt
: > $out
for rep in 1 2; do
  for c in 1 2 4; do
    r=$(DSOPP_HIP_COMB_COPIES=$c timeout 300 python scripts/threshold_sweep.py 7 2000 2>/dev/null | grep "us per" | sed 's/.*: //')
    echo "rep $rep  copies $c  7 KF / 2000: $r" | tee -a $out
  done
done
for c in 1 4; do
  r=$(DSOPP_HIP_COMB_COPIES=$c timeout 300 python scripts/threshold_sweep.py 7 8000 2>/dev/null | grep "us per" | sed 's/.*: //')
  echo "copies $c  7 KF / 8000: $r" | tee -a $out
  r=$(DSOPP_HIP_COMB_COPIES=$c timeout 300 python scripts/threshold_sweep.py 4 1000 2>/dev/null | grep "us per" | sed 's/.*: //')
  echo "copies $c  4 KF / 1000: $r" | tee -a $out
done
for c in 1 4; do
  d=/tmp/prof_c$c; rm -rf $d
  (cd /tmp && DSOPP_HIP_COMB_COPIES=$c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o prof -- python $GRAFT_REPO_ROOT/scripts/profile_target.py c1 > /tmp/prof_c$c.log 2>&1)
  t=$(find $d -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && python scripts/one_solve_timeline.py "$t" > $O/c1_copies${c}_one_solve_timeline.csv && echo "copies $c: $(tail -1 $O/c1_copies${c}_one_solve_timeline.csv)" | tee -a $out
done
timeout 1500 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_degenerate.py tests/test_gpu_masks.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_copies.log
