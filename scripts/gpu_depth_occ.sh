#!/bin/bash
# tuning aid: depth-estimation kernel time for several occupancy targets (DEPTH_WAVES -> __launch_bounds__(64, n))
cd $GRAFT_REPO_ROOT
for n in 0 4 5 6; do
  DSOPP_HIP_EXTRA_FLAGS="-DDSOPP_DEPTH_WAVES=$n" bash dsopp_amd/csrc/build.sh > /dev/null 2>&1
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pd && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o prof -- python $GRAFT_REPO_ROOT/scripts/profile_target.py depth > /tmp/pd.log 2>&1)
  echo "waves $n: $(grep -h "estimateDepthsBatch" $(find /tmp/pd -name "*kernel_stats.csv") | sed 's/.*DepthLandmarks const\*)",//')"
done
bash dsopp_amd/csrc/build.sh > /dev/null 2>&1
