#!/bin/bash
# round-5 measurement pass on the GPU box: what DESIGN.md §6 quotes for the round's FINAL build lands under gpurun_out/r05/ (copied into
# profiles/r05/).  The round's experiments have scripts of their own: gpu_r5_probe.sh (atomics fan-in), gpu_r5_fullres.sh (1280 x 1024:
# parity, bench extra, counter traffic), gpu_r5_order.sh (internal landmark order A/B + large-loop timeline), gpu_r5_concurrent.sh
# (concurrent windows), gpu_r5_solver.sh (window sizes behind the 512-thread solve launch).
#   usage: bash scripts/gpu_round5.sh [quick]     quick: skip the test suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
if [ "$1" != "quick" ]; then
  timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2
fi
# rocprofv3 --kernel-trace --stats per workload (+ one-solve timelines of the loops)
for what in c1 c1_isolated large_loop c3_loop tracker; do
  d=/tmp/prof_$what; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o prof -- python $GRAFT_REPO_ROOT/scripts/profile_target.py $what > $GRAFT_REPO_ROOT/$O/$what.log 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/${what}_kernel_stats.csv || echo "no stats for $what"
  if [ "$what" = c1 ] || [ "$what" = large_loop ] || [ "$what" = c3_loop ]; then
    t=$(find $d -name '*kernel_trace.csv' | head -1)
    [ -n "$t" ] && python scripts/one_solve_timeline.py "$t" > $O/${what}_one_solve_timeline.csv
  fi
  rm -rf $d $O/$what.log
done
# TCC traffic of the C1 kernels (bench.py reads per_launch_bytes from the newest committed file)
timeout 600 python scripts/pmc_traffic.py c1 > $O/pmc_c1.log 2>&1; cp gpurun_out/pmc_traffic.json $O/pmc_traffic_c1.json 2>/dev/null
# ... and of the 12 KF / 50 k window (the library's internal landmark order is what runs)
timeout 900 python scripts/pmc_traffic.py large > $O/pmc_large.log 2>&1; cp gpurun_out/pmc_traffic_large.json $O/pmc_traffic_large.json 2>/dev/null
# bench lines: the driver's form, then the default
python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_steps20.err | grep "^{" > $O/bench_steps20.json
python bench.py 2>$O/bench.err | grep "^{" > $O/bench.json
python - <<'PY'
import json
for f in ("bench", "bench_steps20"):
    try:
        d = json.load(open(f"gpurun_out/r05/{f}.json"))
        print(f, round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 5), "roofline", d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
try:
    d = json.load(open("gpurun_out/r05/bench.json"))
    json.dump(d["roofline_large_fullres"], open("gpurun_out/r05/fullres_windows.json", "w"), indent=1)
except Exception as e:
    print("fullres extra missing", e)
PY
ls $O
