#!/bin/bash
# per-kernel in-loop durations (rocprofv3 --kernel-trace --stats) of the C1 loop for several builds of the library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
for lib in ${LIBS:-lib lib_exp_pro}; do
  for what in ${WHATS:-c1}; do
  d=/tmp/prof_$lib; rm -rf $d
  (cd /tmp && DSOPP_HIP_LIB=$GRAFT_REPO_ROOT/dsopp_amd/$lib/libdsopp_hip.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o prof -- python $GRAFT_REPO_ROOT/scripts/profile_target.py $what > /tmp/prof_$lib.log 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== $lib $what" | tee -a $O/prologue_kernel_stats.txt
  [ -n "$f" ] && head -8 "$f" | cut -d, -f1-4 | tee -a $O/prologue_kernel_stats.txt
  t=$(find $d -name '*kernel_trace.csv' | head -1)
  [ -n "$t" ] && python scripts/one_solve_timeline.py "$t" > $O/${what}_${lib}_one_solve_timeline.csv && tail -1 $O/${what}_${lib}_one_solve_timeline.csv
  done
done
