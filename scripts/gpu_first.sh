set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -m3 -E "gfx|Marketing" 
nproc; lscpu | grep -m1 "Model name"
timeout 900 python -m pytest tests/test_gpu_pba.py -x -q -m gpu 2>&1 | tail -40
