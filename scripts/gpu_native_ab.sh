#!/bin/bash
# same-box A/B of the native tick driver between two builds of the library (the driver links dsopp_amd/lib: the builds are swapped in)
#   usage: bash scripts/gpu_native_ab.sh <other lib dir> [size]
cd $GRAFT_REPO_ROOT
other=$1; size=${2:-1280x1024}
cp -r dsopp_amd/lib /tmp/lib_this
for round in 1 2; do
  for v in this other; do
    rm -rf dsopp_amd/lib; if [ $v = this ]; then cp -r /tmp/lib_this dsopp_amd/lib; else cp -r $other dsopp_amd/lib; fi
    python scripts/time_tick_native.py $size 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', 'frame %.3f keyframe %.3f all %.3f' % (d['ms_per_frame_mean'], d['ms_per_keyframe_mean'], d['ms_per_frame_including_keyframe_work']), {k: round(v,3) for k,v in d['ms_per_frame_by_phase'].items() if v}, {k: round(v,3) for k,v in d['ms_per_keyframe_by_phase'].items()})"
  done
done
rm -rf dsopp_amd/lib; cp -r /tmp/lib_this dsopp_amd/lib
