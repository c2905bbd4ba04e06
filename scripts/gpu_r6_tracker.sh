#!/bin/bash
# round 6: the persistent tracker kernel with the matrix-core workgroup sums against the round-5 kernel (lib_exp_base): parity tests, ms / frame,
# phase stamps of one pass
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_tracker.py tests/test_depth_maps.py tests/test_tracker_hypotheses.py tests/test_gpu_masks.py tests/test_gpu_tick_sequence.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r06/tracker_pytest.log
for lib in lib lib_exp_base; do
  for i in 1 2; do DSOPP_HIP_LIB=$PWD/dsopp_amd/$lib/libdsopp_hip.so python scripts/time_tracker.py 2>/dev/null | tail -1; done
done | tee gpurun_out/r06/tracker_ab.txt
for lib in lib_stamps lib_exp_base_stamps; do
  echo "== $lib"
  DSOPP_HIP_TRACE=1 DSOPP_HIP_LIB=$PWD/dsopp_amd/$lib/libdsopp_hip.so python scripts/time_tracker.py 2>&1 | grep "alignPyramid pass" | sort | uniq -c | sort -rn | head -8
done | tee gpurun_out/r06/tracker_stamps.txt
