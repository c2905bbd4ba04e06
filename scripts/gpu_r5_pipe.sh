#!/bin/bash
# round 5: software-pipelined linearisation sweep (front half of group g + 1 issued under the back half of group g), built into
# dsopp_amd/lib_pipe: parity subset, then A/B against the shipped library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
Q=$GRAFT_REPO_ROOT/dsopp_amd/lib_pipe/libdsopp_hip.so
DSOPP_HIP_LIB=$Q timeout 1500 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_golden.py tests/test_gpu_masks.py tests/test_gpu_invariances.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_pipe.log
for rep in 1 2; do for cfg in "12 50000" "7 20000" "7 2000" "15 5000" "12 8000"; do
  echo "shipped: $(timeout 300 python scripts/time_large.py $cfg 2>&1 | tail -1)"
  echo "pipe:    $(DSOPP_HIP_LIB=$Q timeout 300 python scripts/time_large.py $cfg 2>&1 | tail -1)"
done; done | tee $O/time_pipe_ab.txt
