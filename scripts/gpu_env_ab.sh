#!/bin/bash
# same-box A/B of the native tick driver between two settings of one environment variable:  bash scripts/gpu_env_ab.sh VAR A B [size]
cd $GRAFT_REPO_ROOT
var=$1; a=$2; b=$3; size=${4:-1280x1024}
for round in 1 2; do
  for v in "$a" "$b"; do
    env $var=$v python scripts/time_tick_native.py $size 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$var=$v', 'frame %.3f keyframe %.3f all %.3f' % (d['ms_per_frame_mean'], d['ms_per_keyframe_mean'], d['ms_per_frame_including_keyframe_work']), {k: round(v,3) for k,v in d['ms_per_frame_by_phase'].items() if v}, {k: round(v,3) for k,v in d['ms_per_keyframe_by_phase'].items()})"
  done
done
