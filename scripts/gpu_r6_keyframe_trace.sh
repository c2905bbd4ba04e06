#!/bin/bash
# round 6: where the host time of the native keyframe path goes — the C++ tick driver under rocprofv3 --hip-trace (+ kernels, copies), 80 frames at 1280x1024
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
python scripts/native_trace_prepare.py /tmp/kf_trace 80 2>&1 | tail -1
/tmp/kf_trace/tick_sequence /tmp/kf_trace/sequence.bin /tmp/kf_trace/poses.txt | tail -1 | tee gpurun_out/r06/native_untraced.json | cut -c1-1500
(cd /tmp && DSOPP_TICK_PHASE_LOG=/tmp/kf_trace/phases.txt rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kf_trace/prof -- /tmp/kf_trace/tick_sequence /tmp/kf_trace/sequence.bin /tmp/kf_trace/poses2.txt > /tmp/kf_trace/traced.json 2>/tmp/kf_trace/rocprof.log); tail -1 /tmp/kf_trace/traced.json | cut -c1-300
ls /tmp/kf_trace/prof/* | head
python - <<'PY'
import json, subprocess, sys
d = json.loads(open('/tmp/kf_trace/traced.json').read().strip().splitlines()[-1])
nf, nk = d.get('frames', 80), d.get('keyframes', 1)
r = subprocess.run([sys.executable, 'scripts/hip_api_breakdown.py', '/tmp/kf_trace/prof', str(nf), str(nk), '/tmp/kf_trace/phases.txt'], capture_output=True, text=True)
open('gpurun_out/r06/keyframe_hip_trace_breakdown.json', 'w').write(r.stdout)
j = json.loads(r.stdout)
print(json.dumps(j.get('phase_log')), r.stderr[-2000:])
for ph, v in sorted((j.get('by_phase') or {}).items()):
    print(ph, json.dumps(v)[:1500])
PY
