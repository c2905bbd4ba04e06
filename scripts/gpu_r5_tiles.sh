#!/bin/bash
# round 5: which internal landmark order?  tile edge 16 / 32 / 64 pixels x order inside the tile (raster, Morton, snake), on the randomly
# ordered 12 KF / 50 k window and the 7 KF / 20 k one; and the generator's own tile order left alone (the round-4 figure) in the same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
for rep in 1 2; do
for tb in 4 5 6; do for inner in raster morton snake; do
  env="DSOPP_HIP_LANDMARK_TILE=$tb DSOPP_HIP_LANDMARK_INNER=$inner"
  for cfg in "12 50000" "7 20000"; do echo "$env: $(env $env timeout 300 python scripts/time_large.py $cfg 2>&1 | tail -1)"; done
done; done
echo "caller order, generator emits tile32: $(DSOPP_HIP_LANDMARK_ORDER=caller DSOPP_SYN_ORDER=tile32 timeout 300 python scripts/time_large.py 12 50000 2>&1 | tail -1)"
echo "caller order, generator random: $(DSOPP_HIP_LANDMARK_ORDER=caller timeout 300 python scripts/time_large.py 12 50000 2>&1 | tail -1)"
done | tee $O/time_landmark_tiles.txt
