#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
timeout 600 python bench.py --no-cpu 2>gpurun_out/bench.err | tee gpurun_out/bench_quick.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value', d['value'], 'ms/step', d['ms_per_step'])
for k,v in d['kernels'].items(): print(f'  {k:18s} {v[\"avg_us\"]:8.2f} us x {v[\"launches\"]}')
print(d['roofline'])
print('isolated', d['kernels_isolated_avg_us'])
"
tail -5 gpurun_out/bench.err
