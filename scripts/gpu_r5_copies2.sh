#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
out=$O/time_comb_copies_sizes.txt
: > $out
for w in "7 3000" "7 4000" "7 6000" "7 10000" "5 4000" "8 6000"; do
  for c in 1 2 4; do
    r=$(DSOPP_HIP_COMB_COPIES_MIN_CHUNKS=1 DSOPP_HIP_COMB_COPIES=$c timeout 300 python scripts/threshold_sweep.py $w 2>/dev/null | grep "us per" | sed 's/.*: //')
    echo "copies $c  $w: $r" | tee -a $out
  done
done
