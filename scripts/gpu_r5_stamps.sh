#!/bin/bash
# round 5: where does a sweep workgroup's time go at 12 KF / 50 k with sorted landmarks?  (stamps build) + the concurrent-threads test + smoke()
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
S=$GRAFT_REPO_ROOT/dsopp_amd/lib_stamps/libdsopp_hip.so
for order in random tile32 clump; do echo "== $order"; DSOPP_HIP_LIB=$S DSOPP_SYN_ORDER=$order timeout 300 python scripts/dbg_sweep_large.py 12 50000 2>&1 | grep stamps; done | tee $O/sweep_large_stamps.txt
timeout 600 python -m pytest tests/test_gpu_streams.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
