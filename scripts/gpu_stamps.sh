#!/bin/bash
# tuning aid: phase stamps of the solve / reduction kernels at 12 KF / 50 000 points and at C1 (stamps build), then the normal build
# again and the window parity tests
cd $GRAFT_REPO_ROOT
DSOPP_HIP_EXTRA_FLAGS=-DDSOPP_HIP_STAMPS bash dsopp_amd/csrc/build.sh > /dev/null 2>&1
for i in 1 2; do python scripts/dbg_stamps.py 12 50000 2>/dev/null | grep -v amdgpu; done
python scripts/dbg_stamps.py 7 2000 2>/dev/null | grep -v amdgpu
bash dsopp_amd/csrc/build.sh > /dev/null 2>&1
timeout 900 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_window_group.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
