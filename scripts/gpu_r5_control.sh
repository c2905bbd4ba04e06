#!/bin/bash
# round 5: the control experiment the round-4 review asked for — the quad-per-item sweep (about 30 % fewer instructions per item) AND
# cache-resident texels together: do the sweep's arithmetic side and its gather side overlap, or add?  12 KF / 50 k, caller order kept
# (DSOPP_HIP_LANDMARK_ORDER=caller) so that the generator's order is what the device walks: random / tile32 / clump (every landmark in
# one 48 x 48 window), shipped sweep against dsopp_amd/lib_quad (scripts/probes/sweep_quad_per_item.patch on this tree, branch quad_trial)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
Q=$GRAFT_REPO_ROOT/dsopp_amd/lib_quad/libdsopp_hip.so
for rep in 1 2; do for order in random tile32 clump; do
  echo "shipped, $order: $(DSOPP_HIP_LANDMARK_ORDER=caller DSOPP_SYN_ORDER=$order timeout 300 python scripts/time_large.py 12 50000 2>&1 | tail -1)"
  echo "quad,    $order: $(DSOPP_HIP_LIB=$Q DSOPP_HIP_LANDMARK_ORDER=caller DSOPP_SYN_ORDER=$order timeout 300 python scripts/time_large.py 12 50000 2>&1 | tail -1)"
done; done | tee $O/time_quad_clump_control.txt
# randomised parity run on the final build (new seeds)
timeout 900 python scripts/stress_parity.py 150 505 2>&1 | tail -3 | tee $O/stress_parity_small.txt
timeout 1200 python scripts/stress_parity.py 30 506 big 2>&1 | tail -3 | tee $O/stress_parity_big.txt
