#!/bin/bash
# round 6: the persistent tracker kernel — parity tests, ms / frame against other builds of the library (lib_exp_*), phase stamps of one pass
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
LIBS="${LIBS:-lib lib_exp_base}"
timeout 1500 python -m pytest tests/test_gpu_tracker.py tests/test_depth_maps.py tests/test_tracker_hypotheses.py tests/test_gpu_masks.py tests/test_gpu_tick_sequence.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r06/tracker_pytest.log
for lib in $LIBS; do
  for i in 1 2; do DSOPP_HIP_LIB=$PWD/dsopp_amd/$lib/libdsopp_hip.so python scripts/time_tracker.py 2>/dev/null | tail -1; done
done | tee gpurun_out/r06/tracker_ab.txt
python - <<'PY'
import json
for l in open('gpurun_out/r06/tracker_ab.txt'):
    d = json.loads(l)
    print(d['lib'].split('/')[-2] if '/' in d['lib'] else d['lib'], '5 levels %.4f ms  4 levels %.4f ms  iterations %g / %g  8 hypotheses %.4f ms' % (
        d['5_levels']['ms_per_frame'], d['4_levels']['ms_per_frame'], d['5_levels']['lm_iterations_per_frame'], d['4_levels']['lm_iterations_per_frame'],
        d['5_levels']['relocalisation_8']['eight_per_launch_ms']))
PY
for lib in ${STAMP_LIBS:-lib_stamps}; do
  echo "== $lib"
  DSOPP_HIP_TRACE=1 DSOPP_HIP_LIB=$PWD/dsopp_amd/$lib/libdsopp_hip.so python scripts/time_tracker.py 2>&1 | grep "alignPyramid pass" | sort | uniq -c | sort -rn | head -4
done | tee gpurun_out/r06/tracker_stamps.txt
