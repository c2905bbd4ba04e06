#!/bin/bash
# round 5: the PBA path at 1280 x 1024 — parity test, bench extra, counter traffic (f64 / f32 texels), one-solve timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullres.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_fullres.log
timeout 900 python -c "
import json, time, torch, bench
from dsopp_amd import capi, synthetic as syn
t0=time.time(); r=bench.run_fullres_windows(capi, syn); r['wall_s']=time.time()-t0
print(json.dumps(r, indent=1))
" 2>&1 | grep -v amdgpu.ids | tee $O/fullres_windows.json
for w in fullres fullres_f32; do
  timeout 1200 python scripts/pmc_traffic.py $w > $O/pmc_$w.log 2>&1; tail -30 $O/pmc_$w.log | head -24
  cp gpurun_out/pmc_traffic_$w.json $O/ 2>/dev/null
done
