#!/bin/bash
# round-6 measurement pass on the GPU box: what DESIGN.md §6 quotes for the round's FINAL build lands under gpurun_out/r06/ (copied into
# profiles/r06/).  Other round-6 scripts: gpu_r6_tracker.sh (persistent tracker kernel: parity, ms / frame against other builds, phase
# stamps), gpu_r6_keyframe_trace.sh (native tick driver under rocprofv3 --hip-trace, calls bucketed by keyframe phase), stress_tracker.py.
#   usage: bash scripts/gpu_round6.sh [quick]     quick: skip the test suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
cat BUILD_COMMIT.txt > $O/BUILD_COMMIT.txt 2>/dev/null
if [ "$1" != "quick" ]; then
  timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.log | tail -2 | tee $O/pytest_gpu.log
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
# SQ counters of THIS build (instruction mix, wait cycles, LDS, f64 matrix-core busy cycles): C1 loop, 12 KF / 50 k loop, tracker
PASS_TIMEOUT=120 bash scripts/sq_counters.sh c1 r06/sq_c1 "sweepKernel|reduceSchur|solveCombined" > $O/sq_c1.txt 2>&1
PASS_TIMEOUT=120 bash scripts/sq_counters.sh large_loop r06/sq_large_loop "sweepKernel|schurTwoStage|combineSystem|solveCombined" > $O/sq_large_loop.txt 2>&1
PASS_TIMEOUT=180 bash scripts/sq_counters.sh tracker r06/sq_tracker "alignPyramidKernel|pyramidAllLevels" > $O/sq_tracker.txt 2>&1
python scripts/mfma_utilisation.py $O > $O/mfma_utilisation.log 2>&1
# TCC traffic: C1 kernels (bench.py reads per_launch_bytes from the newest committed file), 12 KF / 50 k, and the same at 1280 x 1024 in f64 and f32 texels
timeout 600 python scripts/pmc_traffic.py c1 > $O/pmc_c1.log 2>&1; cp gpurun_out/pmc_traffic.json $O/pmc_traffic_c1.json 2>/dev/null
timeout 900 python scripts/pmc_traffic.py large > $O/pmc_large.log 2>&1; cp gpurun_out/pmc_traffic_large.json $O/pmc_traffic_large.json 2>/dev/null
for w in fullres fullres_f32; do
  timeout 1200 python scripts/pmc_traffic.py $w > $O/pmc_$w.log 2>&1; cp gpurun_out/pmc_traffic_$w.json $O/ 2>/dev/null
done
# tracker: ms / frame + phase stamps (stamps build) + bitwise reproducibility
for i in 1 2; do python scripts/time_tracker.py 2>/dev/null | tail -1; done > $O/tracker_ms_per_frame.jsonl
if [ -f dsopp_amd/lib_stamps/libdsopp_hip.so ]; then
  DSOPP_HIP_TRACE=1 DSOPP_HIP_LIB=$PWD/dsopp_amd/lib_stamps/libdsopp_hip.so python scripts/time_tracker.py 2>&1 | grep "alignPyramid pass" | sort | uniq -c | sort -rn | head -6 > $O/tracker_stamps.txt
fi
python scripts/stress_tracker.py 1000 2>/dev/null | grep distinct > $O/stress_tracker.txt
# native tick driver (200 frames, both sizes)
python scripts/time_tick_native.py 1280x1024 640x480 2>/dev/null | grep "^{" > $O/tick_native.jsonl
# bench lines: the driver's form, then the default (full result -> bench_extras.json)
python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_steps20.err | grep "^{" > $O/bench_steps20.json; cp bench_extras.json $O/bench_steps20_extras.json
python bench.py 2>$O/bench.err | grep "^{" > $O/bench.json; cp bench_extras.json $O/bench_extras.json
rm -f $O/bench.err $O/bench_steps20.err
python - <<'PY'
import json
for f in ("bench", "bench_steps20"):
    try:
        t = open(f"gpurun_out/r06/{f}.json").read()
        d = json.loads(t)
        print(f, len(t), "bytes", round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 5), "roofline", d.get("roofline", {}).get("frac"), "tracker", d.get("frame_tracking_ms_per_frame_1280x1024"))
    except Exception as e:
        print(f, "FAILED", e)
PY
ls $O
