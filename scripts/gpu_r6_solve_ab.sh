#!/bin/bash
# round 6: the solve launch's panel factorisation with row_newbcast (default) against the v_readlane form (lib_exp_readlane): parity tests, the
# driver's bench form on both, one-solve timelines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_degenerate.py tests/test_gpu_invariances.py tests/test_gpu_window_group.py tests/test_gpu_sliding_window.py tests/test_marginalization.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -4
for lib in lib lib_exp_readlane lib lib_exp_readlane; do
  DSOPP_HIP_LIB=$PWD/dsopp_amd/$lib/libdsopp_hip.so python bench.py --no-extras --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],1), 'GN it/s', round(d['ms_per_step']*1e3,2), 'us per iteration')"
done | tee gpurun_out/r06/solve_panel_ab.txt
for lib in lib lib_exp_readlane; do
  echo "== $lib: isolated kernels (C1, 12 KF / 50 k, 15 KF / 5 k)"
  DSOPP_HIP_LIB=$PWD/dsopp_amd/$lib/libdsopp_hip.so python - <<'PY'
import sys; sys.path.insert(0, '.')
from dsopp_amd import capi, synthetic as syn
for F, P in ((7, 2000), (12, 50000), (15, 5000)):
    win = syn.make_window(F, P, 640, 480, seed=1 if F == 12 else 0)
    g = capi.HipWindow(capi.default_pba_options()); syn.load_window(g, win); g.snapshot(); g.optimize_repeated(14); g.restore()
    import time
    t0 = time.perf_counter(); n, _ = g.optimize_repeated(56); dt = time.perf_counter() - t0
    print(F, P, 'assemble_solve isolated %.2f us' % g.time_kernel('assemble_solve', 100), ' loop %.2f us / iteration' % (dt / n * 1e6))
    g.close()
PY
done 2>&1 | grep -v amdgpu | tee -a gpurun_out/r06/solve_panel_ab.txt
