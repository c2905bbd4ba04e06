#!/bin/bash
# round 5, first call: atomics fan-in probe (geometry of the landmark-major linearisation) + the round-4 build's C1 line on the same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 scripts/probes/bin/atomic_fanin_probe 2>&1 | tee gpurun_out/atomic_fanin_probe.txt
timeout 600 python bench.py --no-cpu --steps 20 --warmup 5 2>gpurun_out/bench.err | tee gpurun_out/bench_r5_start.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'])
for k,v in d['kernels'].items(): print(f'  {k:18s} {v[\"avg_us\"]:8.2f} us x {v[\"launches\"]}')
"
tail -3 gpurun_out/bench.err
