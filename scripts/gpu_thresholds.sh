#!/bin/bash
cd $GRAFT_REPO_ROOT
for P in 5000 8000 12000 20000 30000; do
  for cfg in "96 96" "100000 96" "100000 100000"; do
    set -- $cfg
    DSOPP_HIP_TWO_STAGE_MIN_CHUNKS=$1 DSOPP_HIP_BACKSUB_SPLIT_MIN_CHUNKS=$2 python scripts/threshold_sweep.py 7 $P 2>/dev/null | grep "us per"
  done
done
for cfg in "96 96" "100000 96"; do
  set -- $cfg
  DSOPP_HIP_TWO_STAGE_MIN_CHUNKS=$1 DSOPP_HIP_BACKSUB_SPLIT_MIN_CHUNKS=$2 python scripts/threshold_sweep.py 12 50000 1 2>/dev/null | grep "us per"
done
