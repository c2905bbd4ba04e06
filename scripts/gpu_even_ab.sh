#!/bin/bash
# A/B of the two-stage Schur launch: chunks dealt out evenly over <= 512 workgroups (default) against ceil(n / 512) chunks per workgroup
cd $GRAFT_REPO_ROOT
for cfg in "12 50000" "7 20000" "9 30000" "16 30000" "12 40000" "12 100000"; do
  for ev in 1 0; do
    echo -n "even=$ev  "; DSOPP_HIP_SCHUR_EVEN=$ev timeout 300 python scripts/time_large.py $cfg 2>&1 | grep -v amdgpu | tail -1
  done
done
