#!/bin/bash
# round 5: timing-only experiment libraries (dsopp_amd/lib_exp, WRONG results) against the shipped one
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
Q=$GRAFT_REPO_ROOT/dsopp_amd/${EXPLIB:-lib_exp}/libdsopp_hip.so
for rep in 1 2; do for cfg in "12 50000" "7 20000" "7 2000"; do
  echo "shipped: $(timeout 300 python scripts/time_large.py $cfg 2>&1 | tail -1)"
  echo "exp:     $(DSOPP_HIP_LIB=$Q timeout 300 python scripts/time_large.py $cfg 2>&1 | tail -1)"
done; done | tee $O/time_exp_ab.txt
