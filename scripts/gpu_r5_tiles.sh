#!/bin/bash
# round 5: (1) solve-launch tail change: parity subset + bench digest; (2) which internal landmark order?  tile edge 16 / 32 / 64 pixels x
# order inside the tile (raster, Morton, snake) on the randomly ordered 12 KF / 50 k window, and the generator's own tile order left alone
# (the round-4 figure) in the same box; (3) the round-4 "DPP quad per item" sweep (scripts/probes/sweep_quad_per_item.patch, rebuilt on
# this tree into dsopp_amd/lib_quad/) now that the landmarks are sorted and the arithmetic side binds
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_invariances.py tests/test_golden.py tests/test_gpu_sliding_window.py tests/test_marginalization.py tests/test_gpu_window_group.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_micro2.log
for i in 1 2; do timeout 600 python bench.py 2>/dev/null | tail -1 > $O/bench_micro2_$i.json; python scripts/bench_brief.py $O/bench_micro2_$i.json; done | tee $O/bench_micro2.txt
if [ -f dsopp_amd/lib_quad/libdsopp_hip.so ]; then
  Q=$GRAFT_REPO_ROOT/dsopp_amd/lib_quad/libdsopp_hip.so
  DSOPP_HIP_LIB=$Q timeout 900 python -m pytest tests/test_gpu_pba.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_quad.log
  for rep in 1 2; do for cfg in "12 50000" "7 20000" "7 2000" "15 5000"; do
    echo "shipped: $(timeout 300 python scripts/time_large.py $cfg 2>&1 | tail -1)"
    echo "quad:    $(DSOPP_HIP_LIB=$Q timeout 300 python scripts/time_large.py $cfg 2>&1 | tail -1)"
  done; done | tee $O/time_quad_ab.txt
fi
for rep in 1 2; do
for tb in 4 5 6; do for inner in raster morton snake; do
  env="DSOPP_HIP_LANDMARK_TILE=$tb DSOPP_HIP_LANDMARK_INNER=$inner"
  for cfg in "12 50000"; do echo "$env: $(env $env timeout 300 python scripts/time_large.py $cfg 2>&1 | tail -1)"; done
done; done
echo "caller order, generator emits tile32: $(DSOPP_HIP_LANDMARK_ORDER=caller DSOPP_SYN_ORDER=tile32 timeout 300 python scripts/time_large.py 12 50000 2>&1 | tail -1)"
echo "caller order, generator random: $(DSOPP_HIP_LANDMARK_ORDER=caller timeout 300 python scripts/time_large.py 12 50000 2>&1 | tail -1)"
done | tee $O/time_landmark_tiles.txt
