#!/bin/bash
# one tracked frame of the native driver as a timeline (kernels, copies, blocking calls between the driver's phase stamps), 1280x1024
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
python scripts/native_trace_prepare.py /tmp/kf_trace 80 2>&1 | tail -1
(cd /tmp && DSOPP_TICK_PHASE_LOG=/tmp/kf_trace/phases.txt rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kf_trace/prof -- /tmp/kf_trace/tick_sequence /tmp/kf_trace/sequence.bin /tmp/kf_trace/poses2.txt > /tmp/kf_trace/traced.json 2>/tmp/kf_trace/rocprof.log)
python scripts/frame_timeline.py /tmp/kf_trace/prof /tmp/kf_trace/phases.txt frame > gpurun_out/r06/frame_timeline.txt
python scripts/frame_timeline.py /tmp/kf_trace/prof /tmp/kf_trace/phases.txt keyframe > gpurun_out/r06/keyframe_timeline.txt
tail -5 gpurun_out/r06/frame_timeline.txt
